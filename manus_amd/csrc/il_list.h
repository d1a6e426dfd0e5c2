// The span list of the image loss from precomputed target maps (k_image_loss_list_mapped), as a device function two kernels
// share: the loss library's own launch (image_loss.hip) and, since round 6, extra workgroups of the forward's last kernel
// (k_fwd_items, raster_fwd.hip: mgr_views_forward_attach_loss_list) -- the list needs the forward's tile offsets, not its image,
// and as a launch of its own it was 8 us of a chain of small kernels between the forward blend and the loss.
#ifndef MANUS_IL_LIST_H
#define MANUS_IL_LIST_H
#include "mgr_common.h"

#define IL_T 128                    // threads of k_image_loss
#define IL_ND (2 * IL_T)            // 256 derivative positions, two per thread
#define IL_H1 5                     // halo of the derivative maps
#define IL_W (IL_ND - 2 * IL_H1)    // 246 outputs per workgroup and row
#define ILS_T 256                   // threads of the list kernels
#define ILS_MAXW 16384
#define ILM_R 8                     // row pairs per workgroup of the mapped list

static inline int64_t il_blocks(int V, int H, int W) { return (int64_t)V * ((H + 1) / 2) * ((W + IL_W - 1) / IL_W); }

struct IlListArgs {      // nbx == 0: no list attached
    int H, W, gxb, HP, nbx, list_views;     // image, spans per row pair, row pairs, workgroups per view = ceil(HP / ILM_R), views
    const uint32_t* tmap;                   // target-vs-background column masks (mgr_image_loss_target_map)
    const uint32_t* tile_start;             // the forward's tile-list offsets
    float2* partial;                        // loss workspace: per-span sums | work list | counters (image_loss.hip)
    uint32_t* work_list;
    uint32_t* work_count;
};
static inline IlListArgs il_list_args(int V, int H, int W, const uint32_t* tmap, const uint32_t* tile_start, void* loss_workspace) {
    const int64_t nb = il_blocks(V, H, W);
    IlListArgs a;
    a.H = H; a.W = W; a.gxb = (W + IL_W - 1) / IL_W; a.HP = (H + 1) / 2; a.nbx = (a.HP + ILM_R - 1) / ILM_R; a.list_views = V;
    a.tmap = tmap; a.tile_start = tile_start;
    a.partial = (float2*)loss_workspace;
    a.work_list = (uint32_t*)((char*)loss_workspace + (size_t)nb * sizeof(float2));
    a.work_count = (uint32_t*)((char*)loss_workspace + (size_t)nb * (sizeof(float2) + sizeof(uint32_t)) + 64);
    return a;
}

#ifdef __HIPCC__
// One thread per (row pair, span), ILM_R row pairs per workgroup of ILS_T threads; a workgroup collects its listed spans in
// LDS and takes ONE slot range of the global list.  s2: two words of LDS.
__device__ __forceinline__ void il_list_mapped_block(const IlListArgs& a, int bx, int v, uint32_t* s2) {
    const int tid = threadIdx.x;
    if (tid == 0) s2[0] = 0;
    __syncthreads();
    const int H = a.H, W = a.W, gxb = a.gxb, HP = a.HP;
    const int nwords = (W + 31) / 32;
    const int gxt = (W + 15) / 16, T = gxt * ((H + 15) / 16);
    const int n_items = ILM_R * gxb;
    constexpr int PER = (ILM_R * ((ILS_MAXW + IL_W - 1) / IL_W) + ILS_T - 1) / ILS_T;
    uint32_t my_rank[PER], my_bid[PER];
    int nmine = 0;
    for (int it = tid; it < n_items; it += ILS_T, ++nmine) {
        const int rp = bx * ILM_R + it / gxb, b = it % gxb;
        my_rank[nmine] = 0xFFFFFFFFu;
        my_bid[nmine] = 0;
        if (rp >= HP) continue;
        const int h0 = rp * 2, rows = h0 + 1 < H ? 2 : 1;
        const uint32_t* trow = a.tile_start + (size_t)v * T + (size_t)(h0 >> 4) * gxt;
        const int lo = max(b * IL_W - 2 * IL_H1, 0), hi = min(b * IL_W + IL_ND, W);  // [lo, hi)
        uint32_t any = trow[((hi - 1) >> 4) + 1] == trow[lo >> 4] ? 0u : 1u;   // some tile under the span holds Gaussians
        const uint32_t* mrow = a.tmap + ((size_t)v * HP + rp) * (size_t)nwords;
        for (int k = lo >> 5; k <= (hi - 1) >> 5 && !any; ++k) {
            uint32_t m = mrow[k];
            const int base = k << 5;
            if (lo > base) m &= ~0u << (lo - base);
            if (hi < base + 32) m &= ~0u >> (base + 32 - hi);
            any |= m;
        }
        const uint32_t bid = ((uint32_t)v * (uint32_t)HP + (uint32_t)rp) * (uint32_t)gxb + (uint32_t)b;
        my_bid[nmine] = bid;
        if (any) my_rank[nmine] = atomicAdd(&s2[0], 1u);
        else {
            const int wcnt = min((b + 1) * IL_W, W) - b * IL_W;
            a.partial[bid] = make_float2(0.f, (float)(wcnt * 3 * rows));
        }
    }
    __syncthreads();
    if (tid == 0 && s2[0]) s2[1] = atomicAdd(a.work_count, s2[0]);
    __syncthreads();
    for (int k = 0; k < nmine; ++k)
        if (my_rank[k] != 0xFFFFFFFFu) a.work_list[s2[1] + my_rank[k]] = my_bid[k];
}
#endif
#endif
