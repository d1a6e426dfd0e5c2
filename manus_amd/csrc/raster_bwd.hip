// Rasterizer backward for gfx950.
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians_backward (autograd
// of the call at /root/reference/src/utils/gaussian_utils.py:407-416; algorithm:
// SURVEY.md Appendix A, "Blend backward (K7)" and "Preprocess backward (K8+K9)").
//
// Design: the upstream kernel issues 9 float atomics per (pixel, Gaussian) hit.
// Here the per-Gaussian gradient "scatter" is a deterministic two-step gather:
//   k_blend_bwd   per tile, back to front; the 64 pixels of a wave are reduced
//                 with DPP row operations, the 4 waves of the tile through LDS, and
//                 ONE 48-byte record per (tile, Gaussian) pair is written to the
//                 pair's private slot (slot = contiguous range per Gaussian), tagged
//                 with the call's epoch;
//   k_preprocess_bwd  per Gaussian, sums its own contiguous slots whose tag matches
//                 and immediately applies the conic -> Sigma3D / mean2D -> mean3D
//                 chain, so no per-Gaussian accumulator ever round-trips HBM.
// No float atomics anywhere: images and gradients are bitwise reproducible.
#include <atomic>

#include "instance_math.h"

#ifdef MGR_STATS
__device__ unsigned long long g_stats[16];
extern "C" int mgr_debug_stats(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 16);
}
__device__ unsigned long long g_bhist[3][65];
extern "C" int mgr_debug_bhist(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bhist), sizeof(unsigned long long) * 3 * 65);
}
#define BH(r, c, v) do { const int c_ = (c); const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_bhist[r][c_], v_); } while (0)
#define MGR_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_stats[i], v_); } while (0)
#else
#define MGR_STAT(i, v)
#endif

#ifdef BWD_PROF
// wall_clock64 (100 MHz) per wave and phase of k_blend_bwd: 0 ticket + item record, 1 entry loads + accumulator reset,
// 2 per-quadrant prologue + box test / compaction, 3 pair loop, 5 flush; 6 items, 7 waves
__device__ unsigned long long g_bprof[8];
extern "C" int mgr_debug_bprof(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bprof), sizeof(unsigned long long) * 8);
}
#define BP(k) { const long long now_ = wall_clock64(); acc_[k] += now_ - tp_; tp_ = now_; }
#else
#define BP(k)
#endif

// 1 when any bit of a wave-uniform lane mask is set -- on the scalar unit (written in C the compiler turns the
// comparison into a per-lane select followed by v_readfirstlane)
__device__ __forceinline__ uint32_t mgr_any64(unsigned long long m) {
    uint32_t r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return r;
}

// Work item = (tile, chunk of MGR_CHUNK = 64 list entries) = one WAVE.  The forward pass saved, per pixel, the
// prefix colour and transmittance in front of every chunk, so chunks are independent:
//   T_i        transmittance in front of entry i (forward recurrence from the checkpoint)
//   w_i        = alpha_i * T_i
//   suffix_i   = (output pixel) - (prefix colour through i) = everything behind i incl. background
//   dL/dalpha_i = T_i (c_i . g) - (suffix_i . g) / (1 - alpha_i)
// which is the upstream back-to-front recurrence rearranged.
//
// Persistent waves pull item records from the queue the forward blend appended.  Lane j holds entry j of the chunk
// (index, record, pair slot) for the whole item; the wave then visits the four 8x8 quadrants of the tile one after
// the other (skipping those whose pixels all ended before the chunk): entries are tested against the bounding box of
// the quadrant's pixels that reach the chunk (mgr_box_dead), survivors compacted pairwise into the wave's LDS slab, and
// the per-Gaussian sums over the 64 pixels are formed in two phases per group of 4 pair steps (8 entries):
//   phase 1, lane = pixel: alpha and the sequential transmittance part for two entries per step (packed fp32),
//            leaving (G dL/dalpha, alpha T) of both entries per pixel in an LDS exchange row (four planes, [column][row]);
//   phase 2, lane = (entry, pixel column): the 8 lanes of an entry each run down their column of the quadrant, two rows
//            per step in packed fp32 (dx is
//            constant along a column: only sum v, sum v dy, sum v dy^2 and the three colour sums are accumulated per
//            lane), a transposing reduction over the 8 lanes leaves one of the nine sums in each lane, and the lanes add
//            them to the entry's accumulator row in LDS (same wave, program order: deterministic).
// After the fourth quadrant lane j applies the per-Gaussian factors and writes ONE 48-byte record per (tile, Gaussian)
// pair, tagged with the call's epoch.
//
// Up to round 2 the four quadrants were four waves of a workgroup that met at three barriers per item, and every
// pair step reduced its 18 values across the 64 lanes (45 of ~117 VALU instructions).  Measured per wave (-DBWD_PROF):
// 41 % of the time computing, 25 % waiting for the slowest quadrant, 16 % in the flush, 18 % in the prologue's
// latency.  Here a wave depends on nobody, the list segment and the records are fetched once per item instead of
// once per quadrant, and the cross-lane part of the reduction is 8 lanes wide.
#define BWD_BATCH MGR_CHUNK
// One exchange row per pair step: four planes of 64 floats (v_a, w_a, v_b, w_b; v = G dL/dalpha, w = alpha T).  Pixel
// (column c, row i) sits at 4 c + 32 (i / 4) + i % 4: a phase-2 lane (entry, column) fetches rows 0..3 and 4..7 of its
// column with two ds_read_b128 per plane and adds them up two rows at a time in packed fp32.  Banks (MI355X_MICROARCH,
// LDS): the stores are ds_write_b32, 32 lanes per cycle on (address / 4) mod 32 -- rows 0..3 of the eight columns are 32
// consecutive dwords; a ds_read_b128 serves 16 lanes per cycle on 64 banks, and with planes a and b 128 floats apart and
// rows 288 apart the four (exchange row, entry) x four columns of a lane group take 16 different 16-byte slots.
// (Measured on the way: [column * 8 + row] planes read with ds_read2_b64 cost 29 M conflict cycles of 197 M per launch;
// a conflict-free variant of that, still on ds_read2_b64's 128 B/clk path, 0.4345 ms.)
// BWD_KO (instrumentation only, WRONG RESULTS: cost bounds for experiments, tools/instr/ko_bwd.sh): bit 1 skips phase 2, bit 2 also
// phase 1's exchange stores, bit 4 runs 57 % of the pair steps (what 4x4-pixel boxes would leave: 35.5 % -> 62 % contributing lanes),
// bit 8 replaces phase 2's LDS atomics by plain stores, bit 16 skips phase 1's arithmetic behind the alpha evaluation,
// bit 32 reads ONE pair record line instead of five, bit 64 reads the five lines twice, bit 128 = the LDS atomics of rounds 3-4,
// bit 256 leaves the flush's stores out, bit 512 the per-quadrant pixel-state loads (constants instead), bit 1024 the entry gathers of
// the item prologue (one record for all lanes)
#ifndef BWD_KO
#define BWD_KO 0
#endif
int mgr_bwd_variant_bits(void) {   // (mgr_build_variant, raster_fwd.hip)
    int bits = 0;
    if (BWD_KO != 0) bits |= 1;
#if defined(MGR_STATS) || defined(BWD_PROF)
    bits |= 2;
#endif
#ifdef BWD_WLAST_REDUCE
    bits |= 4;
#endif
#if defined(MGR_GATHER_PRELOAD)
    if (MGR_GATHER_PRELOAD != 1) bits |= 4;      // (set on the command line only by an A/B build: the default is defined further down)
#endif
    return bits;
}
// a volatile load that stays a ds_read (a volatile access through a generic pointer becomes a flat load)
#define MGR_LDS_VOLATILE_F32 volatile const __attribute__((address_space(3))) float*
#define BWD_ROW 288
#define BWD_PLANE_W 64
#define BWD_PLANE_B 128
#ifndef BWD_WAVES
#define BWD_WAVES 4
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BWD_WAVES, BWD_WAVES))) void k_blend_bwd(
    int N, int W, int H, int gx, int gy, const uint32_t* __restrict__ sorted_gid, const MgrGRec* __restrict__ grec,
    const uint32_t* __restrict__ n_contrib, const float4* __restrict__ ckpt, const uint4* __restrict__ items,
    MgrHeader* hdr, const float* __restrict__ out_color, const float* __restrict__ dL_dpix,
    uint32_t* __restrict__ pair_tag, float4* __restrict__ pair_grad, uint32_t* __restrict__ inst_tag, uint32_t cap,
    uint32_t epoch) {
    __shared__ __align__(16) float s_pair[4][32][MGR_PAIR_FLOATS];
    __shared__ __align__(16) float s_xch[4][4 * BWD_ROW];
    __shared__ float s_acc[4][BWD_BATCH][9];
    __shared__ uint32_t s_flag[4][BWD_BATCH];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = gx * gy;
    const uint32_t n_items = hdr->n_items;
    if (blockIdx.x == 0 && tid < 4) (tid == 3 ? hdr->n_active : hdr->n_runs[tid]) = 0;  // for the gather that follows (first view group)
    if (blockIdx.x == 0 && tid == 4) hdr->bwd_seq += 1u;
    const size_t P = (size_t)W * H;
    const unsigned long long lt = (1ull << lane) - 1ull;
    float* const slab = &s_pair[wave][0][0];
    float* const xch = &s_xch[wave][0];
    float* const acc = &s_acc[wave][0][0];
    uint32_t* const flag = &s_flag[wave][0];
    const int e2 = lane >> 3, pc = lane & 7;   // phase-2 role: entry slot of the group (exchange row e2 >> 1, half e2 & 1), pixel column
#ifdef BWD_PROF
    long long acc_[6] = {0, 0, 0, 0, 0, 0}, tp_ = wall_clock64(), nit_ = 0;
#endif

    MgrQueue queue;   // (blocked counters -- the chunks of a tile to one XCD -- measured: 0.426 against 0.422 ms, no gain)
    queue.init(hdr->qctr, n_items, (int)blockIdx.x);
    uint32_t item = queue.resolve(queue.issue(lane), lane);
    while (item != 0xFFFFFFFFu) {
        const uint32_t next_raw = queue.issue(lane);   // the next ticket's round trip hides behind this item
        const uint4 recA = items[2 * (size_t)item], recB = items[2 * (size_t)item + 1];
        const uint32_t vt = recA.x, chunk = recA.y;
        const uint32_t mxq[4] = {recB.x, recB.y, recB.z, recB.w};
        const int v = (int)(vt / (uint32_t)T), t = (int)(vt % (uint32_t)T);
        const int bx = t % gx, by = t / gx;
        const uint32_t first = chunk * BWD_BATCH;             // list position of the chunk's first entry
        const uint32_t tmax = max(max(mxq[0], mxq[1]), max(mxq[2], mxq[3]));
        const int cnt = (int)min((uint32_t)BWD_BATCH, tmax - first);
        const MgrGRec* const gv = grec + (size_t)v * N;
#ifdef BWD_PROF
        ++nit_;
#endif
        BP(0);
        // lane j: entry j of the chunk (clamped index: the loads are unconditional, the results masked)
#if BWD_KO & 1024
        const uint32_t gid = sorted_gid[recA.z] + 0u * (uint32_t)lane;
#else
        const uint32_t gid = sorted_gid[recA.z + (uint32_t)min(lane, cnt - 1)];
#endif
        float4 ra, rb, rcz;   // (x, y, conic A, B | conic C, opacity, colour r, g | colour b, slot base, rect width, -)
        {
            const MgrGRec* r = gv + gid;
            ra = *(const float4*)r;
            rb = *((const float4*)r + 1);
            rcz = *((const float4*)r + 2);
        }
        // accumulator rows of the 64 entries
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k * 64 + lane] = 0.f;
        flag[lane] = 0u;
        const int32_t slot = __float_as_int(rcz.y) + by * __float_as_int(rcz.z) + bx;
        const float qmax_e = mgr_qmax(rb.y);
        BP(1);

        // quadrants with a pixel that reaches the chunk (wave-uniform); the per-pixel state of the next one is fetched
        // while the current one is processed
        uint32_t qmask = (mxq[0] > first ? 1u : 0u) | (mxq[1] > first ? 2u : 0u) | (mxq[2] > first ? 4u : 0u) | (mxq[3] > first ? 8u : 0u);
        struct PxState {
            uint32_t nc;
            float t0, t1, t2, o0, o1, o2;
            float4 ck;
        };
        auto load_px = [&](int quad, PxState& q) {
            const int px = bx * 16 + (quad & 1) * 8 + (lane & 7);
            const int py = by * 16 + (quad >> 1) * 8 + (lane >> 3);
            const size_t pixc = (size_t)min(py, H - 1) * W + min(px, W - 1);
#if BWD_KO & 512
            q.nc = tmax; q.t0 = 0.001f * (float)pixc; q.t1 = 0.5f; q.t2 = 0.25f; q.o0 = 0.3f; q.o1 = 0.2f; q.o2 = 0.1f;
            q.ck = make_float4(0.1f, 0.1f, 0.1f, 0.5f);
#else
            q.nc = n_contrib[(size_t)v * P + pixc];
            const float* gp = dL_dpix + (size_t)v * 3 * P + pixc;
            const float* op = out_color + (size_t)v * 3 * P + pixc;
            q.t0 = gp[0]; q.t1 = gp[P]; q.t2 = gp[2 * P];
            q.o0 = op[0]; q.o1 = op[P]; q.o2 = op[2 * P];
            q.ck = ckpt[(size_t)recA.w * 256 + ((quad << 6) | lane)];
#endif
        };
        PxState pxn;
        load_px(qmask ? __builtin_ctz(qmask) : 0, pxn);
#pragma unroll 1
        while (qmask) {
            const int quad = __builtin_ctz(qmask);
            qmask &= qmask - 1u;
            const PxState pxs = pxn;
            load_px(qmask ? __builtin_ctz(qmask) : quad, pxn);   // (the last quadrant re-reads its own: harmless, never waited for)
            const int px = bx * 16 + (quad & 1) * 8 + (lane & 7);
            const int py = by * 16 + (quad >> 1) * 8 + (lane >> 3);
            const bool inside = px < W && py < H;
            const mgr_v2f fpx2 = {(float)px, (float)px}, fpy2 = {(float)py, (float)py};
            const float qx0 = (float)(bx * 16 + (quad & 1) * 8), qy0 = (float)(by * 16 + (quad >> 1) * 8);
            // per-pixel state in front of the chunk
            float Tr = 1.0f, pg = 0.f, Og = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
            uint32_t last = 0;
            {
                last = inside ? pxs.nc : 0u;
                if (last > first) {
                    g0 = pxs.t0; g1 = pxs.t1; g2 = pxs.t2;
                    Og = pxs.o0 * g0 + pxs.o1 * g1 + pxs.o2 * g2;
                    if (chunk > 0) {
                        Tr = pxs.ck.w;
                        pg = pxs.ck.x * g0 + pxs.ck.y * g1 + pxs.ck.z * g2;
                    }
                }
            }
            // deepest contributor of this quadrant: the forward recorded exactly this maximum (tile_qdone -> the item record),
            // so it is a scalar here -- not six cross-lane steps (ds_bpermute round trips) per quadrant visit
#ifdef BWD_WLAST_REDUCE
            uint32_t wlast = last;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) wlast = max(wlast, (uint32_t)__shfl_xor((int)wlast, d, 64));
#else
            const uint32_t wlast = quad == 0 ? mxq[0] : quad == 1 ? mxq[1] : quad == 2 ? mxq[2] : mxq[3];
#endif
            int bx0, by0, bx1, by1;
            if (!mgr_quad_bbox(__ballot(last > first), bx0, by0, bx1, by1)) continue;
            // dL/dpixel of this lane's phase-2 column (pixels (pc, 0..7) of the quadrant), through the exchange buffer
            mgr_v2f gr0[4], gr1[4], gr2[4];   // rows (2k, 2k + 1) of the column
            const int xoff = 4 * (lane & 7) + 32 * (lane >> 5) + ((lane >> 3) & 3);   // this pixel's place in a plane
            {
                xch[xoff] = g0; xch[64 + xoff] = g1; xch[128 + xoff] = g2;   // three planes, same layout as the exchange rows
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 a4 = *(const float4*)(xch + 4 * pc + 32 * j), b4 = *(const float4*)(xch + 64 + 4 * pc + 32 * j),
                                 c4 = *(const float4*)(xch + 128 + 4 * pc + 32 * j);
                    gr0[2 * j] = mgr_v2f{a4.x, a4.y}; gr0[2 * j + 1] = mgr_v2f{a4.z, a4.w};
                    gr1[2 * j] = mgr_v2f{b4.x, b4.y}; gr1[2 * j + 1] = mgr_v2f{b4.z, b4.w};
                    gr2[2 * j] = mgr_v2f{c4.x, c4.y}; gr2[2 * j + 1] = mgr_v2f{c4.z, c4.w};
                }
                __builtin_amdgcn_wave_barrier();
            }
            const float fx_col = qx0 + (float)pc;
            const mgr_v2f g0v = {g0, g0}, g1v = {g1, g1}, g2v = {g2, g2};
            bool alive = false;
            if (lane < cnt && first + (uint32_t)lane < wlast)  // later entries are deeper than every pixel's last
                alive = !mgr_box_dead(ra.x, ra.y, ra.z, ra.w, rb.x, qmax_e, qx0 + (float)bx0, qy0 + (float)by0,
                                      qx0 + (float)bx1, qy0 + (float)by1);
            const unsigned long long m = __ballot(alive);
            const int na = __popcll(m);
            MGR_STAT(0, __popcll(__ballot(lane < cnt)));   // (entry, quadrant) box tests
            MGR_STAT(1, na);                                 // survivors
            if (alive) {
                const int rank = __popcll(m & lt);
                float* pb = slab + (rank >> 1) * MGR_PAIR_FLOATS;
                mgr_pair_store(pb, rank & 1, ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w, rcz.x,
                               first + (uint32_t)lane + 1u);  // 1-based list position
                if ((na & 1) && rank == na - 1) mgr_pair_pad(pb);
            }
            const int npair = (BWD_KO & 4) ? ((na + 1) >> 1) * 57 / 100 : (na + 1) >> 1;
#ifdef MGR_STATS
            // what finer pixel blocks could save (tools/instr/blend_stats.py): the quadrant's active pixels, and per 4x4 block the
            // entries with a valid pixel in it -- a wave with one entry list per block would run max_b ceil(n_b / 2) pair steps
            MGR_STAT(8, (unsigned long long)npair * __popcll(__ballot(last > first)));
            MGR_STAT(9, npair);
            uint32_t nblk_[4] = {0u, 0u, 0u, 0u};
#endif
            BP(2);
#pragma unroll 1
            for (int p0 = 0; p0 < npair; p0 += 4) {
                const int np = min(4, npair - p0);
                uint32_t tmask = 0;   // bit 2r + h: entry h of exchange row r is valid at some pixel

                // ---- phase 1: lane = pixel ----
                for (int r = 0; r < np; ++r) {
                    const float4* pp = (const float4*)(slab + (p0 + r) * MGR_PAIR_FLOATS);
#if BWD_KO & 32
                    const float4 R0 = pp[0], R1 = R0, R2 = R0, R3 = R0, R4 = make_float4(R0.x, R0.y, __uint_as_float(first + 1u), __uint_as_float(first + 2u));
#elif BWD_KO & 64
                    float4 R0 = pp[0];
                    const float4 R1 = pp[1], R2 = pp[2], R3 = pp[3], R4 = pp[4];
                    {
                        const float4* qq = (const float4*)(slab + ((p0 + r) ^ 1) * MGR_PAIR_FLOATS);
                        const float4 S0 = qq[0], S1 = qq[1], S2 = qq[2], S3 = qq[3], S4 = qq[4];
                        R0.x = __builtin_fmaf(S0.x + S1.y + S2.z + S3.w + S4.x, 0.0f, R0.x);
                    }
#else
                    const float4 R0 = pp[0], R1 = pp[1], R2 = pp[2], R3 = pp[3], R4 = pp[4];
#endif
                    mgr_v2f dx, dy, G, al;
                    bool va, vb;
                    mgr_pair_alpha(R0, R1, R2, fpx2, fpy2, dx, dy, G, al, va, vb);
                    // valid = alpha kept by the forward rule AND at or in front of the pixel's last contributor; the list
                    // position of a skipped alpha is replaced by UINT_MAX so that ONE compare yields the lane mask
                    const uint32_t ka = va ? __float_as_uint(R4.z) : 0xFFFFFFFFu, kb = vb ? __float_as_uint(R4.w) : 0xFFFFFFFFu;
                    va = ka <= last;
                    vb = kb <= last;
                    const unsigned long long ma = __builtin_amdgcn_ballot_w64(va), mb = __builtin_amdgcn_ballot_w64(vb);
                    const uint32_t anya = mgr_any64(ma), anyb = mgr_any64(mb);  // wave-uniform
                    MGR_STAT(2, 1);                                                    // pair iterations
                    MGR_STAT(3, __popcll(__ballot(va)) + __popcll(__ballot(vb)));      // valid (entry, pixel) evaluations
                    MGR_STAT(4, anya + anyb);                      // entries with any valid pixel
#ifdef MGR_STATS
                    {
                        const unsigned long long bm_[4] = {0x0F0F0F0Full, 0xF0F0F0F0ull, 0x0F0F0F0Full << 32, 0xF0F0F0F0ull << 32};
                        uint32_t nb_ = 0;
                        for (int b_ = 0; b_ < 4; ++b_) {
                            const uint32_t ha_ = (ma & bm_[b_]) != 0ull, hb_ = (mb & bm_[b_]) != 0ull;
                            nblk_[b_] += ha_ + hb_;
                            nb_ += ha_ + hb_;
                        }
                        MGR_STAT(6, nb_);                          // (entry, 4x4 block) combinations with a valid pixel
                    }
#endif
                    if ((ma | mb) == 0ull) continue;
                    MGR_STAT(5, 1);                                                    // pair iterations doing the full math
                    tmask |= (anya | (anyb << 1)) << (2 * r);
#if BWD_KO & 16
                    {
                        float* const xr = xch + r * BWD_ROW + xoff;
                        xr[0] = al.x; xr[BWD_PLANE_W] = G.x; xr[BWD_PLANE_B] = al.y; xr[BWD_PLANE_B + BWD_PLANE_W] = G.y;
                        continue;
                    }
#endif
                    const mgr_v2f cr = {R3.x, R3.y}, cgn = {R3.z, R3.w}, cb = {R4.x, R4.y};
                    const mgr_v2f cg = cr * g0v + cgn * g1v + cb * g2v;
                    // both entries side by side; only the transmittance and the prefix chain from a to b
                    const mgr_v2f a2 = {va ? al.x : 0.0f, vb ? al.y : 0.0f};
                    const mgr_v2f oma = 1.0f - a2;
                    const mgr_v2f rc = {__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};
                    const mgr_v2f T2 = {Tr, Tr * oma.x};          // transmittance in front of a, of b
                    Tr = T2.y * oma.y;
                    const mgr_v2f w2 = a2 * T2;
                    const mgr_v2f wc = w2 * cg;
                    mgr_v2f P2;                                   // prefix . g through a, through b
                    P2.x = pg + wc.x;
                    P2.y = P2.x + wc.y;
                    pg = P2.y;
                    const mgr_v2f d2 = T2 * cg - (Og - P2) * rc;
                    const mgr_v2f da2 = {va ? d2.x : 0.0f, vb ? d2.y : 0.0f};
                    const mgr_v2f v_op = G * da2;   // dL/dopacity share; times the opacity = q = dL/dG G (applied in the flush)
#if BWD_KO & 2
                    if (v_op.x + w2.x + v_op.y + w2.y == 12345.678f) xch[xoff] = 1.f;   // (keeps the arithmetic alive)
#else
                    float* const xr = xch + r * BWD_ROW + xoff;
                    xr[0] = v_op.x; xr[BWD_PLANE_W] = w2.x;
                    xr[BWD_PLANE_B] = v_op.y; xr[BWD_PLANE_B + BWD_PLANE_W] = w2.y;
#endif
                }
                if (tmask == 0u) continue;
#if BWD_KO & 3
                if (tmask != 0xFFFFFFFFu) { flag[(p0 * 2) & 63] = 1u; continue; }
#endif
                __builtin_amdgcn_wave_barrier();
                // ---- phase 2: lane = (entry e2, pixel column pc) ----
                // moments of v = G dL/dalpha about the Gaussian's centre; dx is constant along a column
                {
                    const int prow = min(p0 + (e2 >> 1), npair - 1), h = e2 & 1;
                    const float* pbs = slab + prow * MGR_PAIR_FLOATS;
                    const float xe = pbs[0 + h], ye = pbs[2 + h];
                    // (the position picked up in phase 1 instead -- two selects per pair step there -- measured 0.361 against 0.356 ms)
                    const uint32_t pos = __float_as_uint(*(MGR_LDS_VOLATILE_F32)(pbs + 18 + h));
                    const float dxc = xe - fx_col, dy0 = ye - qy0;
                    const float* src = xch + (e2 >> 1) * BWD_ROW + h * BWD_PLANE_B + 4 * pc;
#if !(BWD_KO & 128)
                    // the entry's accumulator row: read now, written back with this quadrant's sums at the end of the step.
                    // The eight entries of a group are distinct list entries, so the 64 + 8 addresses of the step are distinct
                    // and the wave's own program order is all the ordering there is -- no LDS atomic (measured: the
                    // ds_add_f32 pair cost 0.05 of the kernel's 0.41 ms, knock-out BWD_KO=8)
                    // (volatile: issued here, unconditionally, behind the slab reads -- as plain loads the compiler sinks them
                    // into the conditional block at the end of the step, three dependent LDS round trips in a row)
                    const bool on2 = (tmask >> e2) & 1u;
                    float* const accp = acc + (on2 ? (int)((pos - 1u - first) & 63u) : 0) * 9;
                    const float old_t = *(MGR_LDS_VOLATILE_F32)(accp + pc), old_9 = *(MGR_LDS_VOLATILE_F32)(accp + 8);
#endif
                    mgr_v2f DY = {dy0, dy0 - 1.0f};
                    mgr_v2f B0 = {0.f, 0.f}, B1 = {0.f, 0.f}, B2 = {0.f, 0.f}, Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 2; ++j) {   // rows 4j .. 4j + 3, two side by side
                        const float4 v4 = *(const float4*)(src + 32 * j), w4 = *(const float4*)(src + BWD_PLANE_W + 32 * j);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int k = 2 * j + q;
                            const mgr_v2f vv = q ? mgr_v2f{v4.z, v4.w} : mgr_v2f{v4.x, v4.y};
                            const mgr_v2f ww = q ? mgr_v2f{w4.z, w4.w} : mgr_v2f{w4.x, w4.y};
                            const mgr_v2f t1 = vv * DY;
                            B0 += vv;
                            B1 += t1;
                            B2 += t1 * DY;
                            Cr += ww * gr0[k];
                            Cg += ww * gr1[k];
                            Cb += ww * gr2[k];
                            DY -= 2.0f;
                        }
                    }
                    const float A0 = B0.x + B0.y, A1 = B1.x + B1.y, A2 = B2.x + B2.y;
                    const float sr = Cr.x + Cr.y, sgn = Cg.x + Cg.y, sb = Cb.x + Cb.y;
                    const float Sx = dxc * A0;
                    const float x8[8] = {Sx, A1, Sx * dxc, dxc * A1, A2, A0, sr, sgn};
                    const float tot = grp8_reduce_scatter(x8, pc);
                    const float b9 = grp_sum<8>(sb);
                    if ((tmask >> e2) & 1u) {
                        const int ja = (int)(pos - 1u - first);
#if BWD_KO & 8
                        acc[ja * 9 + pc] = tot;
                        if (pc == 0) {
                            acc[ja * 9 + 8] = b9;
                            flag[ja] = 1u;
                        }
#elif BWD_KO & 128
                        __hip_atomic_fetch_add(acc + ja * 9 + pc, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (pc == 0) {
                            __hip_atomic_fetch_add(acc + ja * 9 + 8, b9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            flag[ja] = 1u;
                        }
#else
                        accp[pc] = old_t + tot;
                        if (pc == 0) {
                            accp[8] = old_9 + b9;
                            flag[ja] = 1u;
                        }
#endif
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
#ifdef MGR_STATS
            MGR_STAT(10, (max(max(nblk_[0], nblk_[1]), max(nblk_[2], nblk_[3])) + 1u) >> 1);
            MGR_STAT(11, (nblk_[0] + nblk_[1] + nblk_[2] + nblk_[3] + 7u) >> 3);     // perfectly balanced over the four blocks
#endif
            BP(3);
        }
        // ---- flush: lane j = entry j ----
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt && flag[lane] != 0u && slot >= 0 && (uint32_t)slot < cap) {
            float r[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) r[c] = acc[lane * 9 + c];
            // r = [Sx, Sy, Sxx, Sxy, Syy] of v = G dL/dalpha (q = opacity * v), dL/dopacity, dL/dcolour; the per-Gaussian factors:
            //   dL/dmean2D.x = -W/2 (A Sx + B Sy),  dL/dmean2D.y = -H/2 (C Sy + B Sx),  dL/dconic = -1/2 (Sxx, Sxy, Syy)   (App. A, K7)
#pragma unroll
            for (int c = 0; c < 5; ++c) r[c] *= rb.y;
            const float cA = ra.z, cB = ra.w, cC = rb.x;
            const float mx = (-0.5f * (float)W) * (cA * r[0] + cB * r[1]);
            const float my = (-0.5f * (float)H) * (cC * r[1] + cB * r[0]);
#if BWD_KO & 256
            if (mx + my + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8] == 12345.678f) pair_tag[slot] = epoch;
#else
            float4* o = pair_grad + (size_t)slot * 3;
            o[0] = make_float4(mx, my, -0.5f * r[2], -0.5f * r[3]);
            o[1] = make_float4(-0.5f * r[4], r[5], r[6], r[7]);
            o[2] = make_float4(r[8], 0.f, 0.f, 0.f);
            pair_tag[slot] = epoch;
            inst_tag[(size_t)v * N + gid] = epoch;  // "this (view, Gaussian) has records": lets the gather skip the rest
#endif
        }
        __builtin_amdgcn_wave_barrier();
        BP(5);
        item = queue.resolve(next_raw, lane);
    }
#ifdef BWD_PROF
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&g_bprof[k], (unsigned long long)acc_[k]);
        atomicAdd(&g_bprof[6], (unsigned long long)nit_);
        atomicAdd(&g_bprof[7], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_preprocess_bwd(
    int N, int W, int H, const float* __restrict__ cams, const float* __restrict__ means3D,
    int64_t s_means, const float* __restrict__ cov3D, int64_t s_cov,
    const int32_t* __restrict__ radii_ws, const ushort4* __restrict__ rect,
    const uint32_t* __restrict__ pair_off, const uint32_t* __restrict__ pair_tag,
    const float4* __restrict__ pair_grad, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcov3D, const uint32_t* __restrict__ inst_tag,
    uint32_t cap, uint32_t epoch, MgrHeader* hdr) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < MGR_NCTR) hdr->qctr[threadIdx.x * 64] = 0u;   // (see k_inst_gather)
    const int v = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t vi = (size_t)v * N + i;
    const ushort4 rc = rect[vi];
    const uint32_t cnt = (uint32_t)((rc.z - rc.x) * (rc.w - rc.y));
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dm[3] = {0.f, 0.f, 0.f}, dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cnt > 0 && inst_tag[vi] == epoch) {  // the blend wrote at least one record for this (view, Gaussian)
        gather_pair_grads(pair_off[vi], cnt, pair_tag, pair_grad, cap, epoch, acc);
        MgrCam cam;
        mgr_load_cam(cams, v, cam);
        const float* mp = means3D + (size_t)v * s_means + (size_t)i * 3;
        const float p[3] = {mp[0], mp[1], mp[2]};
        const float* cp = cov3D + (size_t)v * s_cov + (size_t)i * 6;
        const float c6[6] = {cp[0], cp[1], cp[2], cp[3], cp[4], cp[5]};
        project_backward(cam, W, H, p, c6, acc, dm, dc6);
    }
    float* o3 = dL_dmeans3D + vi * 3;
    o3[0] = dm[0]; o3[1] = dm[1]; o3[2] = dm[2];
    float* o2 = dL_dmeans2D + vi * 3;
    o2[0] = acc[0]; o2[1] = acc[1]; o2[2] = 0.f;
    float* oc = dL_dcolors + vi * 3;
    oc[0] = acc[6]; oc[1] = acc[7]; oc[2] = acc[8];
    dL_dopacity[vi] = acc[5];
    float* ov = dL_dcov3D + vi * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) ov[j] = dc6[j];
}


// ---------------------------------------------------------------------------
// Fused per-instance backward, one thread per (Gaussian, view) with the G views of a Gaussian
// in G adjacent lanes (lane = instance_local * G + view_local):
//   * per-Gaussian loads (canonical parameters, skin weights, SH coefficients) are shared by the
//     G lanes of a group, so a wave touches 64/G records instead of 64 per load;
//   * per-view data (camera, bone transforms) sits in LDS, one padded slab per view;
//   * every lane does the whole chain for its view -- pair-gradient gather, projection backward,
//     SH backward, LBS backward -- and the per-view contributions are summed across the group
//     with DPP (fixed tree: deterministic), by a transposing reduce-scatter when G = 8 so that
//     lane k ends up with output k of each block of 8 and the stores of a group are contiguous;
//   * nothing is staged through global memory between the phases.
// `accumulate` adds to the outputs instead of writing them (view groups beyond the first).
// ---------------------------------------------------------------------------
// Every lane of a group of G < 8 holds the same eight totals t[0..7]: lane r stores the 8 / G consecutive ones from r * 8 / G
// with one vector store (dst is only dword-aligned) -- not lane 0 all eight one after the other: with runs of 4 / 2 / 1 lanes
// (k_inst_bwd_runs) that was 8 store instructions of 16 / 32 / 64 scattered dwords per block of eight values.
typedef float mgr_f2u __attribute__((ext_vector_type(2), aligned(4)));
template <int G>
__device__ __forceinline__ void grp_store_span(const float t[8], int vl, bool ok, float* __restrict__ dst, int n, bool accumulate) {
    constexpr int PER = 8 / G;
    float mine[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        mine[j] = t[j];
#pragma unroll
        for (int r = 1; r < G; ++r) mine[j] = vl == r ? t[r * PER + j] : mine[j];
    }
    const int e0 = vl * PER;
    if (!ok || e0 >= n) return;
    float* d = dst + e0;
    if (e0 + PER <= n) {
        if (PER == 2) {
            mgr_f2u v = {mine[0], mine[1]};
            if (accumulate) v += *(const mgr_f2u*)d;
            *(mgr_f2u*)d = v;
        } else if (PER == 4) {
            mgr_f4u v = {mine[0], mine[1], mine[2], mine[3]};
            if (accumulate) v += *(const mgr_f4u*)d;
            *(mgr_f4u*)d = v;
        } else {
#pragma unroll
            for (int h = 0; h < PER / 4; ++h) {
                mgr_f4u v = {mine[4 * h], mine[4 * h + 1], mine[4 * h + 2], mine[4 * h + 3]};
                if (accumulate) v += *(const mgr_f4u*)(d + 4 * h);
                *(mgr_f4u*)(d + 4 * h) = v;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (e0 + j < n) d[j] = accumulate ? d[j] + mine[j] : mine[j];
    }
}

// Sum x[0..7] over the group and store them at dst[0..n) (n <= 8).
template <int G>
__device__ __forceinline__ void grp_store8(const float x[8], int vl, bool ok, float* __restrict__ dst, int n,
                                           bool accumulate) {
    if (G == 8) {
        const float t = grp8_reduce_scatter(x, vl);
        if (ok && vl < n) dst[vl] = accumulate ? dst[vl] + t : t;
    } else {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = grp_sum<G>(x[k]);
        grp_store_span<G>(t, vl, ok, dst, n, accumulate);
    }
}

#ifndef IB_THREADS
#define IB_THREADS 256
#endif
// waves per SIMD of k_inst_bwd: at 3 (168 VGPRs) the kernel spills 57 registers, at 2 (256) none -- 0.158 -> 0.127 ms at eight
// views (4: 0.225); re-measured late in round 3, the round-2 setting was 3
#ifndef MGR_IB_WAVES
#define MGR_IB_WAVES 2
#endif
#ifndef IG_ROUNDS
#define IG_ROUNDS 8
#endif
//  // lane-group rounds per workgroup of the gather (amortises the list append)
#define IB_TSTRIDE(B) ((B) * 16 + 4)  // LDS words per view of bone transforms (+4: the G slabs fall in distinct banks)

// Phase 1: gather.  Sums the pair-gradient records of every (Gaussian, view), keeps the 9 sums
// for phase 2, appends the Gaussians that received anything in this view group to the active
// list, writes the per-Gaussian statistics (visibility count, max radius) and -- unless
// accumulating -- zero gradients for the Gaussians that received nothing.
template <int G>
__global__ __launch_bounds__(256) void k_inst_gather(
    int v_first, int v_count, int N, int B, int n_art, const int32_t* __restrict__ radii, const ushort4* __restrict__ rect,
    const uint32_t* __restrict__ pair_off, const uint32_t* __restrict__ pair_tag,
    const float4* __restrict__ pair_grad, const uint32_t* __restrict__ inst_tag, uint32_t cap, uint32_t epoch,
    int accumulate, int rounds, float4* __restrict__ iacc,
    uint32_t* __restrict__ active_list, MgrHeader* hdr, float* __restrict__ d_xyz, float* __restrict__ d_ls,
    float* __restrict__ d_rot, float* __restrict__ d_op, float* __restrict__ d_fdc, float* __restrict__ d_frest,
    float* __restrict__ d_w, float* __restrict__ st_grad2d, float* __restrict__ st_vis,
    int32_t* __restrict__ st_radii, uint32_t* __restrict__ run_list) {
    constexpr int IPB = 256 / G;
    __shared__ uint32_t s_list[IPB * IG_ROUNDS];
    __shared__ uint32_t s_cnt, s_base;
    // run_list (G = 8 only): besides the active list, the active Gaussians by lane class -- entry = Gaussian | view mask << 24,
    // the mask of the views whose record sums are non-zero; class c = 0 / 1 / 2 holds the Gaussians with 5..8 / 3..4 / 1..2
    // such views = runs of 8 / 4 / 2 lanes of k_inst_bwd_runs.  (Single-lane runs for the Gaussians with one view were
    // measured: 0.105 against 0.099 ms -- 64 Gaussians per wave make every load and store of a row 64 cache lines wide.)
    constexpr int RL = G == 8 ? IPB * IG_ROUNDS : 1;
    __shared__ uint32_t s_rl[3][RL];
    __shared__ uint32_t s_rc[3], s_rb[3];
    const bool runs = G == 8 && run_list != nullptr;
    const int tid = threadIdx.x, vl = tid & (G - 1), il = tid / G, lane = tid & 63;
    if (blockIdx.x == 0 && tid < MGR_NCTR && v_first == 0) hdr->qctr[tid * 64] = 0u;   // the blend's queue is drained: ready for the next backward
    if (tid == 0) s_cnt = 0;
    if (tid < 3) s_rc[tid] = 0;
    __syncthreads();
    const bool acc_out = accumulate != 0;
#pragma unroll 1
    for (int rnd = 0; rnd < rounds; ++rnd) {  // rounds <= IG_ROUNDS (sizes s_list); fewer when that leaves too few workgroups
        const int i_raw = (blockIdx.x * rounds + rnd) * IPB + il;
        const int i = min(i_raw, N - 1);
        const bool ok = i_raw < N, mine = ok && vl < v_count;
        const int v = v_first + vl;
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.f;
        int rad = 0;
        bool any = false;
        {   // the four per-(view, Gaussian) words are fetched together (clamped view index), then masked
            const size_t vi = (size_t)min(v, v_first + v_count - 1) * N + i;
            const int rd = radii[vi];
            const uint32_t tg = inst_tag[vi];
            const ushort4 rc = rect[vi];
            const uint32_t po = pair_off[vi];
            rad = mine ? rd : 0;
            if (rad > 0 && tg == epoch) {
                gather_pair_grads(po, (uint32_t)((rc.z - rc.x) * (rc.w - rc.y)), pair_tag, pair_grad, cap, epoch, acc);
#pragma unroll
                for (int k = 0; k < 9; ++k) any = any || (acc[k] != 0.f);
            }
        }
        const bool grp_any = grp_sum<G>(any ? 1.0f : 0.0f) > 0.0f;
        const float vis = grp_sum<G>(rad > 0 ? 1.0f : 0.0f);
        int maxrad = rad;
        if (G >= 2) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0xb1, 0xf, 0xf, false));
        if (G >= 4) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0x4e, 0xf, 0xf, false));
        if (G >= 8) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0x141, 0xf, 0xf, false));
        if (runs) {
            const uint32_t m8 = (uint32_t)(__ballot(any) >> (lane & 56)) & 0xFFu;   // this Gaussian's views with a non-zero sum
            if (ok && vl == 0 && m8) {
                const int nv = __popc(m8), c = nv > 4 ? 0 : nv > 2 ? 1 : 2;
                s_rl[c][atomicAdd(&s_rc[c], 1u)] = (uint32_t)i | (m8 << 24);
            }
        }
        if (ok && (runs ? any : grp_any)) {  // phase 2 reads these back, lane for lane (the lanes of the view mask, with run lists)
            float4* o = iacc + ((size_t)i * G + vl) * 3;
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            o[2] = make_float4(acc[8], 0.f, 0.f, 0.f);
        }
        {   // active Gaussians of the block are collected in LDS (one LDS atomic per wave and round)
            const bool lead = ok && grp_any && vl == 0;
            const unsigned long long m = __ballot(lead);
            if (m) {
                const int first = __builtin_ctzll(m);
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, first, 64);
                if (lead) s_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
            }
        }
        if (ok && vl == 0) {
            if (st_vis) st_vis[i] = acc_out ? st_vis[i] + vis : vis;
            if (st_radii) st_radii[i] = acc_out ? max(st_radii[i], maxrad) : maxrad;
        }
    }
    if (!acc_out) {
        // Zero gradients for the whole contiguous range of Gaussians this workgroup owns, fully coalesced; phase 2
        // then overwrites the rows of the active ones (it runs after this kernel).  Row-wise zeroing of only the
        // inactive rows was strided per lane and dominated this kernel when few views share a lane group.
        const size_t i_lo = (size_t)blockIdx.x * rounds * IPB;
        const size_t i_hi = min((size_t)N, i_lo + (size_t)rounds * IPB);
        if (i_hi > i_lo) {
            const size_t n = i_hi - i_lo;
            for (size_t k = tid; k < n * 3; k += 256) { d_xyz[i_lo * 3 + k] = 0.f; d_ls[i_lo * 3 + k] = 0.f; d_fdc[i_lo * 3 + k] = 0.f; }
            for (size_t k = tid; k < n * 4; k += 256) d_rot[i_lo * 4 + k] = 0.f;
            for (size_t k = tid; k < n * 45; k += 256) d_frest[i_lo * 45 + k] = 0.f;
            if (d_w && i_lo < (size_t)n_art) {  // skin-weight rows exist for the articulated Gaussians only
                const size_t na = min(i_hi, (size_t)n_art) - i_lo;
                for (size_t k = tid; k < na * (size_t)B; k += 256) d_w[i_lo * B + k] = 0.f;
            }
            for (size_t k = tid; k < n; k += 256) {
                d_op[i_lo + k] = 0.f;
                if (st_grad2d) st_grad2d[i_lo + k] = 0.f;
            }
        }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt;
    if (tid == 0 && cnt) s_base = atomicAdd(&hdr->n_active, cnt);  // one global atomic per block
    __syncthreads();
    for (uint32_t k = tid; k < cnt; k += 256) active_list[s_base + k] = s_list[k];
    if (runs) {
        if (tid < 3) s_rb[tid] = s_rc[tid] ? atomicAdd(&hdr->n_runs[tid], s_rc[tid]) : 0u;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (uint32_t k = tid; k < s_rc[c]; k += 256) run_list[(size_t)c * N + s_rb[c] + k] = s_rl[c][k];
    }
}

// Phase 1 with run lists (k_inst_bwd_runs behind it; G = 1: the active list alone, k_inst_bwd<1> behind it).  k_inst_gather<G> gives every (Gaussian, view) of
// its range a lane that walks the instance's pair slots -- 18 % of those lanes have records on the bench step, and a wave
// lasts as long as its lane with the most records.  Here the workgroup first compacts the instances that hold records
// (radius > 0 and the epoch tag) of its <= 256 Gaussians into an LDS work list (A), then walks the list with all lanes
// busy (B: the record sums of an instance by one lane, in slot order as before: the same values bit for bit), and last
// turns the per-Gaussian view masks the walk left in LDS into the active list and the run lists (C).
template <int G>
__global__ __launch_bounds__(256) void k_inst_gather_runs(
    int v_first, int v_count, int N, int B, int n_art, const int32_t* __restrict__ radii, const ushort4* __restrict__ rect,
    const uint32_t* __restrict__ pair_off, const uint32_t* __restrict__ pair_tag,
    const float4* __restrict__ pair_grad, const uint32_t* __restrict__ inst_tag, uint32_t cap, uint32_t epoch,
    int accumulate, int rounds, float4* __restrict__ iacc,
    uint32_t* __restrict__ active_list, MgrHeader* hdr, float* __restrict__ d_xyz, float* __restrict__ d_ls,
    float* __restrict__ d_rot, float* __restrict__ d_op, float* __restrict__ d_fdc, float* __restrict__ d_frest,
    float* __restrict__ d_w, float* __restrict__ st_grad2d, float* __restrict__ st_vis,
    int32_t* __restrict__ st_radii, uint32_t* __restrict__ run_list, unsigned char* __restrict__ row_state, int kept) {
    // row_state (one byte per Gaussian; single view group only, else nullptr): 1 = the gradient rows of this Gaussian in the
    // caller's buffers were written by the previous backward on this workspace.  kept: the caller vouches that the buffers are
    // those of that call, untouched -- then only the rows that were written then and get nothing now are zeroed (in place of
    // 324 + 4 B bytes for every Gaussian of the range: 97 MB of the bench step's 151 MB of gather writes), provided the row
    // state is the previous call's (hdr->rows_seq + 1 == hdr->bwd_seq; any backward through another path breaks the chain).
    constexpr int IPB = 256 / G, NG = 256;   // at most 256 Gaussians per workgroup (the caller keeps rounds <= G)
    __shared__ uint32_t s_wi[NG * G], s_wo[NG * G], s_wc[NG * G];   // work list: (local Gaussian << 3 | view), first slot, slots
    __shared__ uint32_t s_mask[NG];                                 // per Gaussian: views with a non-zero sum
    __shared__ uint32_t s_list[NG], s_rl[3][NG];
    __shared__ uint32_t s_nw, s_cnt, s_base, s_rc[3], s_rb[3];
    const int tid = threadIdx.x, vl = tid & (G - 1), il = tid / G, lane = tid & 63;
    if (blockIdx.x == 0 && tid < MGR_NCTR && v_first == 0) hdr->qctr[tid * 64] = 0u;   // the blend's queue is drained: ready for the next backward
    if (tid == 0) { s_nw = 0; s_cnt = 0; }
    if (tid < 3) s_rc[tid] = 0;
    s_mask[tid] = 0u;    // (NG == 256 threads)
    __syncthreads();
    const bool acc_out = accumulate != 0;
    const int i_base = blockIdx.x * rounds * IPB;
    const unsigned long long owner = (unsigned long long)(uintptr_t)d_xyz;
    const bool kept_ok = kept && row_state && hdr->rows_seq != 0u && hdr->rows_seq + 1u == hdr->bwd_seq &&
                         hdr->rows_owner[0] == (uint32_t)owner && hdr->rows_owner[1] == (uint32_t)(owner >> 32);
    // ---- A: the instances with records -> work list; visibility statistics
    // (the kernel is ONE round of workgroups -- 1172 of them at 300 k Gaussians -- so its time is a workgroup's chain of dependent
    //  round trips: the four loads of all its rounds are issued together instead of a round trip per round)
    int rd_r[G];
    uint32_t tg_r[G], po_r[G];
    ushort4 rc_r[G];
#ifndef MGR_GATHER_PRELOAD
#define MGR_GATHER_PRELOAD 1
#endif
#if MGR_GATHER_PRELOAD
#pragma unroll
    for (int rnd = 0; rnd < G; ++rnd) {
        const int i = min(i_base + min(rnd, rounds - 1) * IPB + il, N - 1);
        const size_t vi = (size_t)min(v_first + vl, v_first + v_count - 1) * N + i;
        rd_r[rnd] = radii[vi];
        tg_r[rnd] = inst_tag[vi];
        rc_r[rnd] = rect[vi];
        po_r[rnd] = pair_off[vi];
    }
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int rnd = 0; rnd < G; ++rnd) {
        if (rnd >= rounds) break;
        const int i_raw = i_base + rnd * IPB + il;
        const int i = min(i_raw, N - 1);
        const bool ok = i_raw < N, mine = ok && vl < v_count;
#if MGR_GATHER_PRELOAD
        const int rd = rd_r[rnd];
        const uint32_t tg = tg_r[rnd];
        const ushort4 rc = rc_r[rnd];
        const uint32_t po = po_r[rnd];
#else
        const size_t vi = (size_t)min(v_first + vl, v_first + v_count - 1) * N + i;
        const int rd = radii[vi];
        const uint32_t tg = inst_tag[vi];
        const ushort4 rc = rect[vi];
        const uint32_t po = pair_off[vi];
        (void)rd_r; (void)tg_r; (void)rc_r; (void)po_r;
#endif
        const int rad = mine ? rd : 0;
        const bool act = rad > 0 && tg == epoch;
        const float vis = grp_sum<G>(rad > 0 ? 1.0f : 0.0f);
        int maxrad = rad;
        if (G >= 2) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0xb1, 0xf, 0xf, false));
        if (G >= 4) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0x4e, 0xf, 0xf, false));
        if (G >= 8) maxrad = max(maxrad, __builtin_amdgcn_update_dpp(0, maxrad, 0x141, 0xf, 0xf, false));
        const unsigned long long m = __ballot(act);
        if (m) {
            const int first = __builtin_ctzll(m);
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(&s_nw, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, first, 64);
            if (act) {
                const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                s_wi[pos] = ((uint32_t)(rnd * IPB + il) << 3) | (uint32_t)vl;
                s_wo[pos] = po;
                s_wc[pos] = (uint32_t)((rc.z - rc.x) * (rc.w - rc.y));
            }
        }
        if (ok && vl == 0) {
            if (st_vis) st_vis[i] = acc_out ? st_vis[i] + vis : vis;
            if (st_radii) st_radii[i] = acc_out ? max(st_radii[i], maxrad) : maxrad;
        }
    }
    __syncthreads();
    // ---- B: record sums, one instance per lane
    const uint32_t n_work = s_nw;
    for (uint32_t k = (uint32_t)tid; k < n_work; k += 256u) {
        const uint32_t w = s_wi[k];
        float acc[9];
        gather_pair_grads(s_wo[k], s_wc[k], pair_tag, pair_grad, cap, epoch, acc);
        bool any = false;
#pragma unroll
        for (int q = 0; q < 9; ++q) any = any || (acc[q] != 0.f);
        {   // (written for every instance with records, zero sums included: the sums of an instance whose tag is this call's are valid)
            float4* o = iacc + ((size_t)(i_base + (int)(w >> 3)) * G + (w & 7u)) * 3;
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            o[2] = make_float4(acc[8], 0.f, 0.f, 0.f);
        }
        if (any) atomicOr(&s_mask[w >> 3], 1u << (w & 7u));
    }
    if (!acc_out && !kept_ok) {   // zero gradients for the Gaussians of this workgroup (see k_inst_gather)
        const size_t i_lo = (size_t)i_base;
        const size_t i_hi = min((size_t)N, i_lo + (size_t)rounds * IPB);
        if (i_hi > i_lo) {
            const size_t n = i_hi - i_lo;
            for (size_t k = tid; k < n * 3; k += 256) { d_xyz[i_lo * 3 + k] = 0.f; d_ls[i_lo * 3 + k] = 0.f; d_fdc[i_lo * 3 + k] = 0.f; }
            for (size_t k = tid; k < n * 4; k += 256) d_rot[i_lo * 4 + k] = 0.f;
            for (size_t k = tid; k < n * 45; k += 256) d_frest[i_lo * 45 + k] = 0.f;
            if (d_w && i_lo < (size_t)n_art) {
                const size_t na = min(i_hi, (size_t)n_art) - i_lo;
                for (size_t k = tid; k < na * (size_t)B; k += 256) d_w[i_lo * B + k] = 0.f;
            }
            for (size_t k = tid; k < n; k += 256) {
                d_op[i_lo + k] = 0.f;
                if (st_grad2d) st_grad2d[i_lo + k] = 0.f;
            }
        }
    }
    __syncthreads();
    // ---- C: active list and run lists from the view masks
    if (row_state && tid < rounds * IPB && i_base + tid < N) {
        const int i = i_base + tid;
        if (kept_ok && row_state[i] && s_mask[tid] == 0u) {   // written by the previous call, nothing now: this row alone is zeroed
            for (int k = 0; k < 3; ++k) { d_xyz[3 * i + k] = 0.f; d_ls[3 * i + k] = 0.f; d_fdc[3 * i + k] = 0.f; }
            for (int k = 0; k < 4; ++k) d_rot[4 * i + k] = 0.f;
            for (int k = 0; k < 45; ++k) d_frest[(size_t)i * 45 + k] = 0.f;
            if (d_w && i < n_art)
                for (int k = 0; k < B; ++k) d_w[(size_t)i * B + k] = 0.f;
            d_op[i] = 0.f;
            if (st_grad2d) st_grad2d[i] = 0.f;
        }
        row_state[i] = s_mask[tid] != 0u ? 1 : 0;
    }
    if (blockIdx.x == 0 && tid == 0) hdr->rows_pending = row_state ? hdr->bwd_seq : 0u;   // (rows_owner is written by the kernel behind this one:
                                                                                           //  every workgroup here still reads it)
    {
        const uint32_t m8 = tid < rounds * IPB ? s_mask[tid] : 0u;
        if (m8) {
            const uint32_t i = (uint32_t)(i_base + tid);
            s_list[atomicAdd(&s_cnt, 1u)] = i;
            if (G >= 2) {
                const int nv = __popc(m8), c = nv > 4 ? 0 : nv > 2 ? 1 : 2;
                s_rl[c][atomicAdd(&s_rc[c], 1u)] = i | (m8 << 24);
            }
        }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt;
    if (tid == 0 && cnt) s_base = atomicAdd(&hdr->n_active, cnt);
    if (G >= 2 && tid < 3) s_rb[tid] = s_rc[tid] ? atomicAdd(&hdr->n_runs[tid], s_rc[tid]) : 0u;
    __syncthreads();
    for (uint32_t k = tid; k < cnt; k += 256) active_list[s_base + k] = s_list[k];
    if (G >= 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (uint32_t k = tid; k < s_rc[c]; k += 256) run_list[(size_t)c * N + s_rb[c] + k] = s_rl[c][k];
    }
}

// Phase 2: the whole per-view backward chain for the active Gaussians only.
// RUNS: a lane group is a RUN of G lanes for one Gaussian's views that hold records (list entry = Gaussian | view mask << 24,
// lane r of the run takes the r-th set bit, lanes beyond the popcount idle), not its G views -- at eight views 41 % of the
// (active Gaussian, view) lanes had a record (tools/instr/lane_stats.py); runs rounded up to 8 / 4 / 2 lanes take 0.53 of the
// lanes (k_inst_bwd_runs): k_inst_bwd 0.124 -> 0.099 ms -- not 0.53 of it: per Gaussian the rows loaded and stored stay the same.
template <int G, int BMAX, bool MIXED, bool SH_HALF, bool RUNS, int NVV = 8>
__device__ __forceinline__ void inst_bwd_body(
    int blk, int n_active, const uint32_t* __restrict__ active_list,
    int v_first, int v_count, int N, int B, int n_art, int W, int H, const float* __restrict__ cams,
    const float* __restrict__ xyz, const float* __restrict__ log_scale, const float* __restrict__ rot,
    const float* __restrict__ op_logit, const float* __restrict__ f_dc, const float* __restrict__ f_rest,
    const float* __restrict__ skin_w, const float* __restrict__ transforms, const float4* __restrict__ iacc,
    float grad2d_scale, int accumulate,
    float* __restrict__ d_xyz, float* __restrict__ d_ls, float* __restrict__ d_rot, float* __restrict__ d_op,
    float* __restrict__ d_fdc, float* __restrict__ d_frest, float* __restrict__ d_w, float* __restrict__ st_grad2d) {
    constexpr int IPB = IB_THREADS / G;
    constexpr int NV = RUNS ? NVV : G;          // view slabs in LDS, lanes per Gaussian of iacc (NVV: the views per group of a launch with run lists)
    extern __shared__ __align__(16) float s_view[];  // NV x (camera 40 | transforms IB_TSTRIDE(B))
    if (blk * IPB >= n_active) return;
    const int tid = threadIdx.x, vl = tid & (G - 1), il = tid / G;
    const int q = blk * IPB + il;
    const bool ok = q < n_active;             // lane's Gaussian exists (all lanes stay for the DPP sums)
    const uint32_t ent = active_list[min(q, n_active - 1)];
    const int i = RUNS ? (int)(ent & 0xFFFFFFu) : (int)ent;
    int view = vl;                            // this lane's view within the group
    bool lane_on = true;
    if (RUNS) {
        uint32_t m = ent >> 24;
        lane_on = vl < __popc(m);
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
            if (k < vl) m &= m - 1u;            // drop the vl lowest set bits
        view = lane_on ? __builtin_ctz(m | 0x100u) : 0;
    }
    const bool any_tf = skin_w != nullptr;          // workgroup-uniform: the pose slabs are staged in LDS
    // this lane's Gaussian is articulated (uniform over its lane group; over the launch unless MIXED = composite)
    const bool has_tf = any_tf && (!MIXED || i < n_art);
    const int tstride = IB_TSTRIDE(B), vstride = MGR_CAM_FLOATS + (any_tf ? tstride : 0);
    for (int k = tid; k < NV * MGR_CAM_FLOATS; k += IB_THREADS) {
        const int g = k / MGR_CAM_FLOATS, e = k % MGR_CAM_FLOATS;
        s_view[g * vstride + e] = g < v_count ? cams[(size_t)(v_first + g) * MGR_CAM_FLOATS + e] : 0.f;
    }
    if (any_tf)
        for (int k = tid; k < NV * B * 16; k += IB_THREADS) {
            const int g = k / (B * 16), e = k % (B * 16);
            s_view[g * vstride + MGR_CAM_FLOATS + e] = g < v_count ? transforms[(size_t)(v_first + g) * B * 16 + e] : 0.f;
        }
    __syncthreads();
    const float* const Tp = s_view + view * vstride + MGR_CAM_FLOATS;

    float acc[9];
    bool any = false;
    {
        const float4* o = iacc + ((size_t)i * NV + view) * 3;
        const float4 a = o[0], b = o[1], c = o[2];
        acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; acc[3] = a.w;
        acc[4] = b.x; acc[5] = b.y; acc[6] = b.z; acc[7] = b.w;
        acc[8] = c.x;
#pragma unroll
        for (int k = 0; k < 9; ++k) any = any || (acc[k] != 0.f);
        any = any && ok && lane_on && view < v_count;
    }
    GaussCano g;
    cano_load(xyz, log_scale, rot, i, g);
    const bool acc_out = accumulate != 0;
    // The chain is cut in two so that the 48 SH sums are reduced and stored before the LBS part
    // needs its registers (the DPP sums cannot sit inside the divergent branch).
    float tf[12], dm[3], dc6[6], gt[12], dxyz[3] = {0.f, 0.f, 0.f};
    {
        float dsh[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) dsh[k] = 0.f;
        if (any) {  // culled, or hidden behind saturated pixels everywhere, otherwise
            MgrCam cam;
            {
                const float* p = s_view + view * vstride;
                cam.tanfovx = p[0];
                cam.tanfovy = p[1];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    cam.view[k] = p[2 + k];
                    cam.proj[k] = p[18 + k];
                }
                cam.campos[0] = p[34]; cam.campos[1] = p[35]; cam.campos[2] = p[36];
            }
            float p3[3], c6[6];
            blend_tf(has_tf ? skin_w + (size_t)i * B : nullptr, Tp, B, tf);
            lbs_apply(tf, g, p3, c6);
            project_backward(cam, W, H, p3, c6, acc, dm, dc6);
            const float gc[3] = {acc[6], acc[7], acc[8]};
            ShDir D;
            if (has_tf) sh_dir_xyz<true>(g.x, g.y, g.z, tf, cam.campos, D);
            else sh_dir_xyz<false>(g.x, g.y, g.z, tf, cam.campos, D);
#pragma unroll
            for (int k = 0; k < 12; ++k) gt[k] = 0.f;
            if (SH_HALF) {   // f_rest stored as fp16 rows of 48
                const ShCoefMemH c = {f_dc + (size_t)i * 3, (const __half*)f_rest + (size_t)i * MGR_SH_HALF_ROW};
                sh_backward_view(c, D, has_tf, gc, dsh, dxyz, gt);
            } else {
                const ShCoefMem c = {f_dc + (size_t)i * 3, f_rest + (size_t)i * 45};
                sh_backward_view(c, D, has_tf, gc, dsh, dxyz, gt);
            }
        }
        // SH coefficient gradient: 48 floats = f_dc (3) | f_rest (45)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            if (G == 8) {
                const float t = grp8_reduce_scatter(dsh + 8 * m, vl);
                const int e = 8 * m + vl;
                float* dst = e < 3 ? d_fdc + (size_t)i * 3 + e : d_frest + (size_t)i * 45 + (e - 3);
                if (ok) *dst = acc_out ? *dst + t : t;
            } else if (m == 0) {   // the block that straddles f_dc | f_rest: element by element
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = 8 * m + k;
                    const float t = grp_sum<G>(dsh[e]);
                    float* dst = e < 3 ? d_fdc + (size_t)i * 3 + e : d_frest + (size_t)i * 45 + (e - 3);
                    if (ok && vl == 0) *dst = acc_out ? *dst + t : t;
                }
            } else {
                float t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = grp_sum<G>(dsh[8 * m + k]);
                grp_store_span<G>(t, vl, ok, d_frest + (size_t)i * 45 + (8 * m - 3), 8, acc_out);
            }
        }
    }
    float ds[3] = {0.f, 0.f, 0.f};
    float dR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dw[BMAX];
    float dop = 0.f, g2 = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) dw[b] = 0.f;
    if (any) {
        float dtf[12];
        lbs_backward_view<true>(tf, g, dm, dc6, gt, dxyz, ds, dR, dtf);
        if (has_tf) {
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (b < B) {
                    const float* T = Tp + b * 16;
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 12; ++k) a += dtf[k] * T[k];
                    dw[b] = a;
                }
            }
        }
        dop = acc[5];
        g2 = sqrtf(acc[0] * acc[0] + acc[1] * acc[1]) * grad2d_scale;
    }
    if (any_tf && d_w) {  // wave-uniform branch around the DPP sums; only articulated rows own a d_w row
#pragma unroll
        for (int m = 0; m < BMAX / 8; ++m)
            grp_store8<G>(dw + 8 * m, vl, ok && has_tf, d_w + (size_t)i * B + 8 * m, B - 8 * m, acc_out);
    }
    {
        const float sg = 1.0f / (1.0f + expf(-op_logit[i]));
        // xyz (3) | d/dlog s (3) | opacity logit | screen-space gradient norm
        const float o8[8] = {dxyz[0], dxyz[1], dxyz[2], ds[0] * g.s[0], ds[1] * g.s[1], ds[2] * g.s[2],
                             dop * sg * (1.0f - sg), g2};
        float t8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t8[k] = grp_sum<G>(o8[k]);
        float tR[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) tR[k] = grp_sum<G>(dR[k]);
        if (ok && vl == 0) {
            float drot[4];
            quat_backward(g, tR, drot);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d_xyz[3 * i + k] = acc_out ? d_xyz[3 * i + k] + t8[k] : t8[k];
                d_ls[3 * i + k] = acc_out ? d_ls[3 * i + k] + t8[3 + k] : t8[3 + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) d_rot[4 * i + k] = acc_out ? d_rot[4 * i + k] + drot[k] : drot[k];
            d_op[i] = acc_out ? d_op[i] + t8[6] : t8[6];
            if (st_grad2d) st_grad2d[i] = acc_out ? st_grad2d[i] + t8[7] : t8[7];
        }
    }
}

#define MGR_IB_PARAMS                                                                                                 \
    int v_first, int v_count, int N, int B, int n_art, int W, int H, const float *__restrict__ cams,                    \
        const float *__restrict__ xyz, const float *__restrict__ log_scale, const float *__restrict__ rot,             \
        const float *__restrict__ op_logit, const float *__restrict__ f_dc, const float *__restrict__ f_rest,          \
        const float *__restrict__ skin_w, const float *__restrict__ transforms, const float4 *__restrict__ iacc,       \
        const uint32_t *__restrict__ active_list, const MgrHeader *__restrict__ hdr, float grad2d_scale, int accumulate, \
        float *__restrict__ d_xyz, float *__restrict__ d_ls, float *__restrict__ d_rot, float *__restrict__ d_op,      \
        float *__restrict__ d_fdc, float *__restrict__ d_frest, float *__restrict__ d_w, float *__restrict__ st_grad2d
#define MGR_IB_ARGS                                                                                                   \
    v_first, v_count, N, B, n_art, W, H, cams, xyz, log_scale, rot, op_logit, f_dc, f_rest, skin_w, transforms, iacc,  \
        grad2d_scale, accumulate, d_xyz, d_ls, d_rot, d_op, d_fdc, d_frest, d_w, st_grad2d
// lane group = the G views of an active Gaussian (active_list: Gaussian indices)
template <int G, int BMAX, bool MIXED, bool SH_HALF>
__global__ __launch_bounds__(IB_THREADS) __attribute__((amdgpu_waves_per_eu(MGR_IB_WAVES, MGR_IB_WAVES))) void k_inst_bwd(MGR_IB_PARAMS) {
    inst_bwd_body<G, BMAX, MIXED, SH_HALF, false>((int)blockIdx.x, (int)hdr->n_active, active_list, MGR_IB_ARGS);
}
// lane group = a run of 8 / 4 / 2 lanes (active_list: the three run lists of k_inst_gather, N entries apart): the
// workgroups of the three classes follow each other in one launch, so that the classes share the rounds of the launch
// (NVV = views per group of the launch: 8, 4 or 2 -- with 4 there are the classes of 4 and 2 lanes, with 2 the one of 2)
template <int NVV, int BMAX, bool MIXED, bool SH_HALF>
__global__ __launch_bounds__(IB_THREADS) __attribute__((amdgpu_waves_per_eu(MGR_IB_WAVES, MGR_IB_WAVES))) void k_inst_bwd_runs(MGR_IB_PARAMS) {
    const int n8 = NVV >= 8 ? (int)hdr->n_runs[0] : 0, n4 = NVV >= 4 ? (int)hdr->n_runs[1] : 0, n2 = (int)hdr->n_runs[2];
    int blk = (int)blockIdx.x;
    if (blk == 0 && threadIdx.x == 0 && hdr->rows_pending == hdr->bwd_seq) {   // the gather left the row state of this call: from
        MgrHeader* h = const_cast<MgrHeader*>(hdr);                              // here on the outputs hold exactly those rows
        const unsigned long long owner = (unsigned long long)(uintptr_t)d_xyz;
        h->rows_seq = hdr->bwd_seq;
        h->rows_owner[0] = (uint32_t)owner;
        h->rows_owner[1] = (uint32_t)(owner >> 32);
    }
    const int b8 = (n8 + IB_THREADS / 8 - 1) / (IB_THREADS / 8), b4 = (n4 + IB_THREADS / 4 - 1) / (IB_THREADS / 4);
    if constexpr (NVV >= 8) {
        if (blk < b8) { inst_bwd_body<8, BMAX, MIXED, SH_HALF, true, NVV>(blk, n8, active_list, MGR_IB_ARGS); return; }
    }
    blk -= b8;
    if constexpr (NVV >= 4) {
        if (blk < b4) { inst_bwd_body<4, BMAX, MIXED, SH_HALF, true, NVV>(blk, n4, active_list + (size_t)N, MGR_IB_ARGS); return; }
    }
    blk -= b4;
    inst_bwd_body<2, BMAX, MIXED, SH_HALF, true, NVV>(blk, n2, active_list + 2 * (size_t)N, MGR_IB_ARGS);
}

// process-wide switch of the run lists (default on; MANUS_INST_RUNS=0 in the environment starts with them off)
static std::atomic<int> g_inst_runs{[] { const char* e = getenv("MANUS_INST_RUNS"); return e ? atoi(e) : 1; }()};
extern "C" int mgr_views_backward_run_lists(int on) { return g_inst_runs.exchange(on); }   // (2: run lists from k_inst_gather<8>, for A/B)

struct CanonGrads {  // fused articulated backward: canonical inputs and leaf-gradient outputs
    int B, n_art, sh_half;
    const float *xyz, *log_scale, *rot, *op_logit, *f_dc, *f_rest, *skin_w, *transforms;
    const int32_t* radii;
    float grad2d_scale;
    float *d_xyz, *d_ls, *d_rot, *d_op, *d_fdc, *d_frest, *d_w, *st_grad2d, *st_vis;
    int32_t* st_radii;
};

static int raster_backward_impl(int V, int N, int W, int H, const float* cams, const float* bg,
                                   const float* means3D, int64_t s_means, const float* cov3D,
                                   int64_t s_cov, const float* colors, int64_t s_col,
                                   const float* opacity, int64_t s_op, const float* out_color,
                                   const float* dL_dcolor, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                                   float* dL_dopacity, float* dL_dcov3D, const CanonGrads* canon, void* workspace,
                                   size_t workspace_bytes, int64_t cap, int debug, void* stream_) {
    (void)colors; (void)s_col; (void)opacity; (void)s_op;  // captured in the workspace records
    hipStream_t stream = (hipStream_t)stream_;
    if (V <= 0 || N < 0 || W <= 0 || H <= 0 || cap < 0)
        return mgr_fail(MGR_EINVAL, "mgr_raster_backward: bad sizes");
    if (!cams || !bg || !dL_dcolor || !out_color || !workspace)
        return mgr_fail(MGR_EINVAL, "mgr_raster_backward: null pointer");
    if (N == 0) return MGR_OK;
    if (!canon && (!means3D || !cov3D || !dL_dmeans3D || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dcov3D))
        return mgr_fail(MGR_EINVAL, "mgr_raster_backward: null pointer");
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    const MgrLayout L = mgr_layout(V, N, W, H, cap);
    if (workspace_bytes < L.total) return mgr_fail(MGR_ENOMEM, "mgr_raster_backward: workspace too small");
    char* ws = (char*)workspace;
    MgrHeader* hdr = (MgrHeader*)(ws + L.header);
    // Pair-gradient records are validated by an epoch tag instead of clearing
    // pair_capacity * 48 bytes per call.  The epoch is a process-wide counter, so a
    // workspace (zero-filled at allocation, tag 0 = never written) never sees the
    // same non-zero value twice (until a 2^32 wrap).
    static std::atomic<uint32_t> g_epoch{0};
    uint32_t epoch = g_epoch.fetch_add(1) + 1;
    if (epoch == 0) epoch = g_epoch.fetch_add(1) + 1;

    // (the queue counters of the blend are left zero by the kernel that follows it, k_inst_gather / k_preprocess_bwd: no
    // memset per call)
    { MGR_PROF("k_blend_bwd", stream); hipLaunchKernelGGL(k_blend_bwd, dim3(256 * BWD_WAVES), dim3(256), 0, stream, N, W, H, gx, gy,
                       (const uint32_t*)(ws + L.sorted_gid), (const MgrGRec*)(ws + L.grec), (const uint32_t*)(ws + L.n_contrib),
                       (const float4*)(ws + L.ckpt), (const uint4*)(ws + L.items), hdr, out_color,
                       dL_dcolor, (uint32_t*)(ws + L.pair_tag), (float4*)(ws + L.pair_grad),
                       (uint32_t*)(ws + L.inst_tag), (uint32_t)cap, epoch); }
    MGR_LAUNCH_CHECK("k_blend_bwd", stream, debug & 1);
    if (canon) {
        // views of a Gaussian share a lane group; more than 8 views go in groups of 8, the later
        // groups adding to the outputs of the first
        const int Gv = V <= 1 ? 1 : V <= 2 ? 2 : V <= 4 ? 4 : 8;
        const int ipb = 256 / Gv;
        const size_t lds = (size_t)Gv * (MGR_CAM_FLOATS + (canon->skin_w ? IB_TSTRIDE(canon->B) : 0)) * sizeof(float);
        float4* iacc = (float4*)(ws + L.inst_grad);
        uint32_t* alist = (uint32_t*)(ws + L.inst_grad + (size_t)N * Gv * 48);
        const int ipb2 = IB_THREADS / Gv;
        const bool mixed = canon->skin_w && canon->n_art < N;
        // gather rounds per workgroup: as many as IG_ROUNDS (amortises the list append) while >= ~1024 workgroups remain
        int rounds = (int)((long long)N / ((long long)ipb * 1024));
        rounds = rounds < 1 ? 1 : (rounds > IG_ROUNDS ? IG_ROUNDS : rounds);
        const dim3 grid((N + ipb2 - 1) / ipb2), grid_g((N + ipb * rounds - 1) / (ipb * rounds));
        for (int v0 = 0; v0 < V; v0 += Gv) {
            const int vc = V - v0 < Gv ? V - v0 : Gv;
            const int accm = v0 > 0 ? 1 : 0;
            if (v0 > 0) {   // the first group's zeros come from k_blend_bwd
                MGR_HIP(hipMemsetAsync(&hdr->n_active, 0, 4, stream));
                MGR_HIP(hipMemsetAsync(&hdr->n_runs[0], 0, 12, stream));
            }
            // run lists (see k_inst_gather_runs, k_inst_bwd_runs): up to 24 transforms, Gaussian indices of 24 bits
            // (at one and two views per group the plain kernels are as fast or faster -- 0.537 against 0.544 ms per step at one view:
            // there every instance with records is a lane group of its own -- so the run lists start at groups of four)
            const bool runs = g_inst_runs.load(std::memory_order_relaxed) != 0 && Gv >= 4 && canon->B <= 24 && N < (1 << 24);
            const int rounds_r = rounds < Gv ? rounds : Gv;   // (at most 256 Gaussians per workgroup of k_inst_gather_runs)
            const dim3 grid_gr((N + ipb * rounds_r - 1) / (ipb * rounds_r));
            uint32_t* rlist = alist + (size_t)N;   // three lists of N entries behind the active list
            // row state of the "outputs kept" mode (debug bit 512): one byte per Gaussian behind the run lists, single view group only
            unsigned char* row_state = V <= Gv ? (unsigned char*)(rlist + 3 * (size_t)N) : nullptr;
            const int kept = (debug & 512) && V <= Gv ? 1 : 0;
#define MGR_IG_LAUNCH(GG)                                                                                             \
    hipLaunchKernelGGL((k_inst_gather<GG>), grid_g, dim3(256), 0, stream, v0, vc, N, canon->B, canon->n_art, canon->radii, \
                       (const ushort4*)(ws + L.rect), (const uint32_t*)(ws + L.pair_off),                             \
                       (const uint32_t*)(ws + L.pair_tag), (const float4*)(ws + L.pair_grad),                        \
                       (const uint32_t*)(ws + L.inst_tag), (uint32_t)cap, epoch, accm, rounds, iacc, alist, hdr, canon->d_xyz, canon->d_ls, canon->d_rot, canon->d_op, canon->d_fdc,    \
                       canon->d_frest, canon->d_w, canon->st_grad2d, canon->st_vis, canon->st_radii, runs && GG == 8 ? rlist : (uint32_t*)nullptr)
#define MGR_IB_LAUNCH(GG, BB)                                                                                         \
    if (mixed && canon->sh_half) MGR_IB_LAUNCH2(GG, BB, true, true);                                                  \
    else if (mixed) MGR_IB_LAUNCH2(GG, BB, true, false);                                                              \
    else if (canon->sh_half) MGR_IB_LAUNCH2(GG, BB, false, true);                                                     \
    else MGR_IB_LAUNCH2(GG, BB, false, false)
#define MGR_IB_LAUNCH2(GG, BB, MX, HF)                                                                                \
    hipLaunchKernelGGL((k_inst_bwd<GG, BB, MX, HF>), grid, dim3(IB_THREADS), lds, stream, v0, vc, N, canon->B, canon->n_art, W, H, cams, canon->xyz, \
                       canon->log_scale, canon->rot, canon->op_logit, canon->f_dc, canon->f_rest, canon->skin_w,      \
                       canon->transforms, (const float4*)iacc, (const uint32_t*)alist, (const MgrHeader*)hdr,         \
                       canon->grad2d_scale, accm, canon->d_xyz, canon->d_ls, canon->d_rot, canon->d_op, canon->d_fdc, \
                       canon->d_frest, canon->d_w, canon->st_grad2d)
            {
                MGR_PROF("k_inst_gather", stream);
#define MGR_IGR_LAUNCH(GG)                                                                                            \
    hipLaunchKernelGGL((k_inst_gather_runs<GG>), grid_gr, dim3(256), 0, stream, v0, vc, N, canon->B, canon->n_art, canon->radii, \
                       (const ushort4*)(ws + L.rect), (const uint32_t*)(ws + L.pair_off),                             \
                       (const uint32_t*)(ws + L.pair_tag), (const float4*)(ws + L.pair_grad),                        \
                       (const uint32_t*)(ws + L.inst_tag), (uint32_t)cap, epoch, accm, rounds_r, iacc, alist, hdr, canon->d_xyz, canon->d_ls, canon->d_rot, canon->d_op, canon->d_fdc, \
                       canon->d_frest, canon->d_w, canon->st_grad2d, canon->st_vis, canon->st_radii, rlist, row_state, kept)
                const bool old_gather = g_inst_runs.load(std::memory_order_relaxed) == 2 && Gv == 8;   // (A/B: run lists from k_inst_gather<8>)
                if (runs && !old_gather) {
                    if (Gv == 8) MGR_IGR_LAUNCH(8);
                    else if (Gv == 4) MGR_IGR_LAUNCH(4);
                    else if (Gv == 2) MGR_IGR_LAUNCH(2);
                    else MGR_IGR_LAUNCH(1);
                }
#undef MGR_IGR_LAUNCH
                else if (Gv == 8) MGR_IG_LAUNCH(8);
                else if (Gv == 4) MGR_IG_LAUNCH(4);
                else if (Gv == 2) MGR_IG_LAUNCH(2);
                else MGR_IG_LAUNCH(1);
            }
            MGR_PROF("k_inst_bwd", stream);
#define MGR_IBR_LAUNCH(MX, HF)                                                                                        \
    if (Gv == 8) MGR_IBR_LAUNCH2(8, MX, HF); else if (Gv == 4) MGR_IBR_LAUNCH2(4, MX, HF); else MGR_IBR_LAUNCH2(2, MX, HF)
#define MGR_IBR_LAUNCH2(NN, MX, HF)                                                                                   \
    hipLaunchKernelGGL((k_inst_bwd_runs<NN, 24, MX, HF>), dim3(grid.x + 3), dim3(IB_THREADS), lds, stream, v0, vc, N, canon->B, canon->n_art, W, H, cams, canon->xyz, \
                       canon->log_scale, canon->rot, canon->op_logit, canon->f_dc, canon->f_rest, canon->skin_w,      \
                       canon->transforms, (const float4*)iacc, (const uint32_t*)rlist, (const MgrHeader*)hdr,         \
                       canon->grad2d_scale, accm, canon->d_xyz, canon->d_ls, canon->d_rot, canon->d_op, canon->d_fdc, \
                       canon->d_frest, canon->d_w, canon->st_grad2d)
            if (runs && Gv >= 2) {
                if (mixed && canon->sh_half) MGR_IBR_LAUNCH(true, true);
                else if (mixed) MGR_IBR_LAUNCH(true, false);
                else if (canon->sh_half) MGR_IBR_LAUNCH(false, true);
                else MGR_IBR_LAUNCH(false, false);
            } else
#undef MGR_IBR_LAUNCH
#undef MGR_IBR_LAUNCH2
            if (canon->B <= 24) {
                if (Gv == 8) MGR_IB_LAUNCH(8, 24);
                else if (Gv == 4) MGR_IB_LAUNCH(4, 24);
                else if (Gv == 2) MGR_IB_LAUNCH(2, 24);
                else MGR_IB_LAUNCH(1, 24);
            } else {
                if (Gv == 8) MGR_IB_LAUNCH(8, MGR_MAX_BONES);
                else if (Gv == 4) MGR_IB_LAUNCH(4, MGR_MAX_BONES);
                else if (Gv == 2) MGR_IB_LAUNCH(2, MGR_MAX_BONES);
                else MGR_IB_LAUNCH(1, MGR_MAX_BONES);
            }
#undef MGR_IB_LAUNCH
#undef MGR_IB_LAUNCH2
#undef MGR_IG_LAUNCH
        }
    } else
    { MGR_PROF("k_preprocess_bwd", stream); hipLaunchKernelGGL(k_preprocess_bwd, dim3((N + 255) / 256, V), dim3(256), 0, stream, N, W, H, cams,
                       means3D, s_means, cov3D, s_cov, (const int32_t*)nullptr,
                       (const ushort4*)(ws + L.rect), (const uint32_t*)(ws + L.pair_off),
                       (const uint32_t*)(ws + L.pair_tag), (const float4*)(ws + L.pair_grad),
                       dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dcov3D,
                       (const uint32_t*)(ws + L.inst_tag), (uint32_t)cap, epoch, hdr); }
    MGR_LAUNCH_CHECK("k_preprocess_bwd", stream, debug & 1);
    return MGR_OK;
}

// Device pointers to the list of Gaussians that received a gradient in the most recent mgr_views_backward on this
// workspace and to its length (valid for V <= 8: one view group; more views leave the last group's list).
extern "C" int mgr_views_active_list(void* workspace, int V, int N, int W, int H, int64_t cap, const uint32_t** list,
                                     const uint32_t** count) {
    if (!workspace || !list || !count || V <= 0 || N <= 0) return mgr_fail(MGR_EINVAL, "mgr_views_active_list: bad arguments");
    if (V > 8) return mgr_fail(MGR_EINVAL, "mgr_views_active_list: only defined for up to 8 views (one lane group)");
    const MgrLayout L = mgr_layout(V, N, W, H, cap);
    const int Gv = V <= 1 ? 1 : V <= 2 ? 2 : V <= 4 ? 4 : 8;
    char* ws = (char*)workspace;
    *list = (const uint32_t*)(ws + L.inst_grad + (size_t)N * Gv * 48);
    *count = &((const MgrHeader*)(ws + L.header))->n_active;
    return MGR_OK;
}

extern "C" int mgr_raster_backward(int V, int N, int W, int H, const float* cams, const float* bg,
                                   const float* means3D, int64_t s_means, const float* cov3D,
                                   int64_t s_cov, const float* colors, int64_t s_col,
                                   const float* opacity, int64_t s_op, const float* out_color,
                                   const float* dL_dcolor, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                                   float* dL_dopacity, float* dL_dcov3D, void* workspace,
                                   size_t workspace_bytes, int64_t cap, int debug, void* stream_) {
    return raster_backward_impl(V, N, W, H, cams, bg, means3D, s_means, cov3D, s_cov, colors, s_col, opacity, s_op,
                                out_color, dL_dcolor, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dcov3D,
                                nullptr, workspace, workspace_bytes, cap, debug, stream_);
}

extern "C" int mgr_views_backward(int V, int N, int B, int n_articulated, int sh_half, int W, int H, const float* cams, const float* bg,
                                  const float* xyz, const float* log_scale, const float* rot,
                                  const float* opacity_logit, const float* f_dc, const float* f_rest,
                                  const float* skin_w, const float* transforms, const int32_t* radii,
                                  const float* out_color, const float* dL_dcolor, float grad2d_scale,
                                  float* d_xyz, float* d_log_scale, float* d_rot, float* d_opacity_logit,
                                  float* d_f_dc, float* d_f_rest, float* d_skin_w, float* stat_grad2d,
                                  float* stat_vis, int32_t* stat_radii, void* workspace, size_t workspace_bytes,
                                  int64_t cap, int debug, void* stream_) {
    if (N > 0 && (!xyz || !log_scale || !rot || !opacity_logit || !f_dc || !f_rest || !radii || !d_xyz ||
                  !d_log_scale || !d_rot || !d_opacity_logit || !d_f_dc || !d_f_rest ||
                  (skin_w && (!transforms || !d_skin_w))))
        return mgr_fail(MGR_EINVAL, "mgr_views_backward: null pointer");
    if (skin_w && (B <= 0 || B > MGR_MAX_BONES)) return mgr_fail(MGR_EINVAL, "mgr_views_backward: bad B");
    if (skin_w && (n_articulated < 0 || n_articulated > N)) return mgr_fail(MGR_EINVAL, "mgr_views_backward: bad n_articulated");
    const CanonGrads cg = {B, skin_w ? n_articulated : 0, sh_half ? 1 : 0, xyz, log_scale, rot, opacity_logit, f_dc, f_rest, skin_w, transforms, radii,
                           grad2d_scale, d_xyz, d_log_scale, d_rot, d_opacity_logit, d_f_dc, d_f_rest, d_skin_w,
                           stat_grad2d, stat_vis, stat_radii};
    return raster_backward_impl(V, N, W, H, cams, bg, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, out_color,
                                dL_dcolor, nullptr, nullptr, nullptr, nullptr, nullptr, &cg, workspace, workspace_bytes,
                                cap, debug, stream_);
}
