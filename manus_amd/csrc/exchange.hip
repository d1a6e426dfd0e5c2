// Row-compacted gradient exchange of the view-sharded training step (manus_amd/engine.py, ViewShardedStep(compact=True)).
//
// No reference counterpart: MANUS trains on one GPU (/root/reference/main.py:84-87, DDP commented out).  With the views of
// a step sharded over the ranks, only the Gaussians that received a gradient on SOME rank need to travel (43 % of the rows
// of the bench scene with all 8 views on one rank).  The step buffer is
//     [ nseg - 1 leaf-gradient segments (N x w_k floats each) | padding | grad2d N | vis N | loss | overflow ]
// and the exchange is
//   mgr_exchange_mask     bytes [ row mask N | visibility count N ] for the small SUM all-reduce that forms the union
//   mgr_exchange_index    ordered list of the union's rows + its length (device; the host reads the length: it sizes the
//                         second collective)
//   mgr_exchange_pack     the union's rows, segment by segment, + (loss, overflow) into the buffer that is all-reduced
//   mgr_exchange_unpack   the reduced rows back into the step buffer (rows outside the union are zero on every rank and
//                         stay untouched), the reduced visibility counts back as floats
// Round 6: mgr_exchange_pack_rows / _unpack_rows take the row count FROM THE DEVICE (the word mgr_exchange_index wrote) and a
// row capacity the host chose beforehand (last step's count + headroom): the second collective is sized without the host
// reading anything in the middle of the step.  Rows between the count and the capacity travel as zeros; a count beyond the
// capacity raises the buffer's overflow word (summed over the ranks like the rasterizer's: the step is run again).
#include "mgr_common.h"

#define XCH_MAX_SEG 8
struct XchSegs {
    int n;
    int width[XCH_MAX_SEG];      // floats per row
    int col0[XCH_MAX_SEG];       // first packed column of the segment
    long long off[XCH_MAX_SEG];  // offset of the segment in the step buffer (floats)
};

static int xch_segs(int nseg, const int64_t* offs, const int* widths, XchSegs& s, const char* who) {
    if (nseg <= 0 || nseg > XCH_MAX_SEG || !offs || !widths) return mgr_fail(MGR_EINVAL, "%s: bad segment table", who);
    s.n = nseg;
    int c = 0;
    for (int k = 0; k < nseg; ++k) {
        if (widths[k] <= 0 || offs[k] < 0) return mgr_fail(MGR_EINVAL, "%s: bad segment table", who);
        s.width[k] = widths[k];
        s.off[k] = offs[k];
        s.col0[k] = c;
        c += widths[k];
    }
    return MGR_OK;
}

// mask from the rows themselves: any non-zero entry in any segment (exact by construction: a row outside the mask is zero)
__global__ __launch_bounds__(256) void k_xch_mask_scan(int N, const float* __restrict__ flat, XchSegs s,
                                                       const float* __restrict__ vis, uint8_t* __restrict__ small) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    bool any = false;
    for (int k = 0; k < s.n; ++k) {
        const float* r = flat + s.off[k] + (size_t)i * s.width[k];
        for (int c = 0; c < s.width[k]; ++c) any = any || (r[c] != 0.0f);   // (NaN != 0: a non-finite row travels)
    }
    small[i] = any ? 1 : 0;
    small[N + i] = (uint8_t)vis[i];
}

// mask from the fused backward's list of the Gaussians that received a gradient (a superset of the non-zero rows; the
// mask bytes were zeroed by the caller's memset)
__global__ __launch_bounds__(256) void k_xch_mask_list(int N, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                       const float* __restrict__ vis, uint8_t* __restrict__ small) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    small[N + i] = (uint8_t)vis[i];
    if ((uint32_t)i < *count) {
        const uint32_t g = list[i];
        if (g < (uint32_t)N) small[g] = 1;
    }
}

#define XCH_BLOCK 1024
__device__ __forceinline__ uint32_t xch_block_scan(uint32_t val, uint32_t* s_w, uint32_t& total) {   // blockDim.x = 1024
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < 16; ++w) { const uint32_t t = s_w[w]; s_w[w] = run; run += t; }
        s_w[16] = run;
    }
    __syncthreads();
    total = s_w[16];
    return s_w[wave] + incl - val;
}

__global__ __launch_bounds__(XCH_BLOCK) void k_xch_count(int N, const uint8_t* __restrict__ mask, uint32_t* __restrict__ part) {
    __shared__ uint32_t s_w[17];
    const int i = blockIdx.x * XCH_BLOCK + threadIdx.x;
    uint32_t total;
    (void)xch_block_scan((i < N && mask[i]) ? 1u : 0u, s_w, total);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

__global__ __launch_bounds__(XCH_BLOCK) void k_xch_index(int N, int nblk, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ part,
                                                         uint32_t* __restrict__ idx, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_w[17];
    __shared__ uint32_t s_base;
    const int i = blockIdx.x * XCH_BLOCK + threadIdx.x;
    // base of this block: the blocks in front of it (a few hundred numbers)
    uint32_t mine = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += XCH_BLOCK) mine += part[j];
    uint32_t tot;
    (void)xch_block_scan(mine, s_w, tot);
    if (threadIdx.x == 0) s_base = tot;
    __syncthreads();
    const bool on = i < N && mask[i];
    const uint32_t r = xch_block_scan(on ? 1u : 0u, s_w, tot);
    if (on) idx[s_base + r] = (uint32_t)i;
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) *count = s_base + tot;
}

template <bool PACK>
__global__ __launch_bounds__(256) void k_xch_move(int N, int n, int cols, const uint32_t* __restrict__ idx, float* __restrict__ flat,
                                                  XchSegs s, long long tail_off, float* __restrict__ buf,
                                                  const uint8_t* __restrict__ small_vis, long long vis_off) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < (long long)n * cols) {
        const int r = (int)(t / cols), c = (int)(t % cols);
        int k = 0;
        while (k + 1 < s.n && c >= s.col0[k + 1]) ++k;
        float* src = flat + s.off[k] + (size_t)idx[r] * s.width[k] + (c - s.col0[k]);
        float* dst = buf + (size_t)n * s.col0[k] + (size_t)r * s.width[k] + (c - s.col0[k]);
        if (PACK) *dst = *src; else *src = *dst;
    }
    if (t < 2) {   // (loss, overflow)
        if (PACK) buf[(size_t)n * cols + t] = flat[tail_off + t];
        else flat[tail_off + t] = buf[(size_t)n * cols + t];
    }
    if (!PACK && small_vis && t < N) flat[vis_off + t] = (float)small_vis[t];
}

template <bool PACK>
__global__ __launch_bounds__(256) void k_xch_move_rows(int N, int cap_rows, const uint32_t* __restrict__ count, int cols,
                                                       const uint32_t* __restrict__ idx, float* __restrict__ flat, XchSegs s, long long tail_off,
                                                       float* __restrict__ buf, const uint8_t* __restrict__ small_vis, long long vis_off) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t cnt = *count;
    const int n = (int)min(cnt, (uint32_t)cap_rows);
    if (t < (long long)cap_rows * cols) {
        const int r = (int)(t / cols), c = (int)(t % cols);
        int k = 0;
        while (k + 1 < s.n && c >= s.col0[k + 1]) ++k;
        float* dst = buf + (size_t)cap_rows * s.col0[k] + (size_t)r * s.width[k] + (c - s.col0[k]);
        if (r < n) {
            float* src = flat + s.off[k] + (size_t)idx[r] * s.width[k] + (c - s.col0[k]);
            if (PACK) *dst = *src; else *src = *dst;
        } else if (PACK) {
            *dst = 0.0f;                               // padding rows: zeros on every rank
        }
    }
    if (t < 2) {   // (loss, overflow): a union larger than the capacity adds to the overflow word
        if (PACK) buf[(size_t)cap_rows * cols + t] = flat[tail_off + t] + ((t == 1 && cnt > (uint32_t)cap_rows) ? 1.0f : 0.0f);
        else flat[tail_off + t] = buf[(size_t)cap_rows * cols + t];
    }
    if (!PACK && small_vis && t < N) flat[vis_off + t] = (float)small_vis[t];
}

static int xch_move_rows(bool pack, int N, int cap_rows, const uint32_t* count, const uint32_t* idx, float* flat, int nseg, const int64_t* offs,
                         const int* widths, int64_t tail_off, float* buf, const uint8_t* small_vis, int64_t vis_off, hipStream_t stream) {
    const char* who = pack ? "mgr_exchange_pack_rows" : "mgr_exchange_unpack_rows";
    if (N < 0 || cap_rows < 0 || cap_rows > N || !count || !flat || !buf || (cap_rows > 0 && !idx) || tail_off < 0)
        return mgr_fail(MGR_EINVAL, "%s: bad arguments", who);
    XchSegs s;
    const int rc = xch_segs(nseg, offs, widths, s, who);
    if (rc != MGR_OK) return rc;
    const int cols = s.col0[nseg - 1] + s.width[nseg - 1];
    long long threads = (long long)cap_rows * cols;
    if (threads < 2) threads = 2;
    if (!pack && small_vis && threads < N) threads = N;
    MGR_PROF(pack ? "k_xch_pack" : "k_xch_unpack", stream);
    if (pack)
        hipLaunchKernelGGL((k_xch_move_rows<true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, N, cap_rows, count, cols, idx, flat, s,
                           (long long)tail_off, buf, (const uint8_t*)nullptr, 0ll);
    else
        hipLaunchKernelGGL((k_xch_move_rows<false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, N, cap_rows, count, cols, idx, flat, s,
                           (long long)tail_off, buf, small_vis, (long long)vis_off);
    MGR_LAUNCH_CHECK(who, stream, 0);
    return MGR_OK;
}

extern "C" int mgr_exchange_pack_rows(int N, int cap_rows, const uint32_t* count, const uint32_t* idx, const float* flat, int nseg, const int64_t* offs,
                                      const int* widths, int64_t tail_off, float* buf, void* stream) {
    return xch_move_rows(true, N, cap_rows, count, idx, const_cast<float*>(flat), nseg, offs, widths, tail_off, buf, nullptr, 0, (hipStream_t)stream);
}

extern "C" int mgr_exchange_unpack_rows(int N, int cap_rows, const uint32_t* count, const uint32_t* idx, float* flat, int nseg, const int64_t* offs,
                                        const int* widths, int64_t tail_off, const float* buf, const uint8_t* small_vis, int64_t vis_off, void* stream) {
    return xch_move_rows(false, N, cap_rows, count, idx, flat, nseg, offs, widths, tail_off, const_cast<float*>(buf), small_vis, vis_off,
                         (hipStream_t)stream);
}

extern "C" int mgr_exchange_mask(int N, const float* flat, int nseg, const int64_t* offs, const int* widths, int64_t vis_off,
                                 const uint32_t* active_list, const uint32_t* active_count, uint8_t* small, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || !small || (N > 0 && !flat) || vis_off < 0) return mgr_fail(MGR_EINVAL, "mgr_exchange_mask: bad arguments");
    if ((active_list == nullptr) != (active_count == nullptr)) return mgr_fail(MGR_EINVAL, "mgr_exchange_mask: list and count go together");
    if (N == 0) return MGR_OK;
    XchSegs s;
    const int rc = xch_segs(nseg, offs, widths, s, "mgr_exchange_mask");
    if (rc != MGR_OK) return rc;
    if (active_list) {
        MGR_HIP(hipMemsetAsync(small, 0, (size_t)N, stream));
        MGR_PROF("k_xch_mask", stream);
        hipLaunchKernelGGL(k_xch_mask_list, dim3((N + 255) / 256), dim3(256), 0, stream, N, active_list, active_count, flat + vis_off, small);
    } else {
        MGR_PROF("k_xch_mask", stream);
        hipLaunchKernelGGL(k_xch_mask_scan, dim3((N + 255) / 256), dim3(256), 0, stream, N, flat, s, flat + vis_off, small);
    }
    MGR_LAUNCH_CHECK("k_xch_mask", stream, 0);
    return MGR_OK;
}

extern "C" size_t mgr_exchange_index_workspace_bytes(int N) { return ((size_t)(N > 0 ? N : 1) + XCH_BLOCK - 1) / XCH_BLOCK * 4 + 16; }

extern "C" int mgr_exchange_index(int N, const uint8_t* mask, uint32_t* idx, uint32_t* count, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || !count || (N > 0 && (!mask || !idx || !workspace))) return mgr_fail(MGR_EINVAL, "mgr_exchange_index: bad arguments");
    if (workspace_bytes < mgr_exchange_index_workspace_bytes(N)) return mgr_fail(MGR_ENOMEM, "mgr_exchange_index: workspace too small");
    if (N == 0) { MGR_HIP(hipMemsetAsync(count, 0, 4, stream)); return MGR_OK; }
    const int nblk = (N + XCH_BLOCK - 1) / XCH_BLOCK;
    uint32_t* part = (uint32_t*)workspace;
    { MGR_PROF("k_xch_count", stream); hipLaunchKernelGGL(k_xch_count, dim3(nblk), dim3(XCH_BLOCK), 0, stream, N, mask, part); }
    { MGR_PROF("k_xch_index", stream); hipLaunchKernelGGL(k_xch_index, dim3(nblk), dim3(XCH_BLOCK), 0, stream, N, nblk, mask, (const uint32_t*)part, idx, count); }
    MGR_LAUNCH_CHECK("k_xch_index", stream, 0);
    return MGR_OK;
}

static int xch_move(bool pack, int N, int n, const uint32_t* idx, float* flat, int nseg, const int64_t* offs, const int* widths,
                    int64_t tail_off, float* buf, const uint8_t* small_vis, int64_t vis_off, hipStream_t stream) {
    const char* who = pack ? "mgr_exchange_pack" : "mgr_exchange_unpack";
    if (N < 0 || n < 0 || n > N || !flat || !buf || (n > 0 && !idx) || tail_off < 0) return mgr_fail(MGR_EINVAL, "%s: bad arguments", who);
    XchSegs s;
    const int rc = xch_segs(nseg, offs, widths, s, who);
    if (rc != MGR_OK) return rc;
    const int cols = s.col0[nseg - 1] + s.width[nseg - 1];
    long long threads = (long long)n * cols;
    if (threads < 2) threads = 2;
    if (!pack && small_vis && threads < N) threads = N;
    MGR_PROF(pack ? "k_xch_pack" : "k_xch_unpack", stream);
    if (pack)
        hipLaunchKernelGGL((k_xch_move<true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, N, n, cols, idx, flat, s,
                           (long long)tail_off, buf, (const uint8_t*)nullptr, 0ll);
    else
        hipLaunchKernelGGL((k_xch_move<false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, N, n, cols, idx, flat, s,
                           (long long)tail_off, buf, small_vis, (long long)vis_off);
    MGR_LAUNCH_CHECK(who, stream, 0);
    return MGR_OK;
}

extern "C" int mgr_exchange_pack(int N, int n, const uint32_t* idx, const float* flat, int nseg, const int64_t* offs, const int* widths,
                                 int64_t tail_off, float* buf, void* stream) {
    return xch_move(true, N, n, idx, const_cast<float*>(flat), nseg, offs, widths, tail_off, buf, nullptr, 0, (hipStream_t)stream);
}

extern "C" int mgr_exchange_unpack(int N, int n, const uint32_t* idx, float* flat, int nseg, const int64_t* offs, const int* widths,
                                   int64_t tail_off, const float* buf, const uint8_t* small_vis, int64_t vis_off, void* stream) {
    return xch_move(false, N, n, idx, flat, nseg, offs, widths, tail_off, const_cast<float*>(buf), small_vis, vis_off, (hipStream_t)stream);
}
