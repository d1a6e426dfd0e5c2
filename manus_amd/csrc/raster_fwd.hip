// Rasterizer forward for gfx950: preprocess -> tile scan/queue -> depth-ordered binning (or: emit ->
// per-tile LDS sort) -> alpha blend.  All V views of a call are batched into every launch.
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians as called from
// /root/reference/src/utils/gaussian_utils.py:393-416 (algorithm: SURVEY.md App. A).
//
// Design (MI355X-first, not the upstream CUDA layout):
//  * no global 64-bit radix sort of the (Gaussian, tile) pairs, and by default no sort of the pairs at all:
//    the instances of a view are sorted once by the unique key (depth_bits << 32 | gaussian_index) and the
//    pairs are scattered to the tile lists in that order ("depth-ordered binning", K3'/K4' below), which
//    reproduces the upstream stable (tile|depth) order exactly.  The round-1 route remains (MGR_BINNING=sorted
//    and very large tile grids): pairs bucketed straight into their tile's segment (block-aggregated LDS
//    histograms -> one global atomic per (block,tile)), then each tile sorted on its own in LDS by that key;
//  * per-(view,Gaussian) data the blend kernels gather is one 48-byte record;
//  * no host synchronisation: the pair count stays on the device, capacity
//    overflow raises a flag in the workspace header.
#include <atomic>
#include <mutex>

#include "instance_math.h"
#include "il_list.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

thread_local char g_mgr_err[512] = {0};

// ---------------------------------------------------------------------------
// event profiler
// ---------------------------------------------------------------------------
int g_mgr_prof_on = 0;
namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void mgr_prof_begin(const char* name, hipStream_t stream) {
    ProfRec r{name, prof_event(), prof_event()};
    if (r.a) (void)hipEventRecord(r.a, stream);
    g_prof_recs.push_back(r);
}
void mgr_prof_end(hipStream_t stream) {
    if (!g_prof_recs.empty() && g_prof_recs.back().b) (void)hipEventRecord(g_prof_recs.back().b, stream);
}

static std::string g_prof_filter;
static int g_prof_period = 1, g_prof_seen = 0;
bool mgr_prof_match(const char* name) {
    if (!(g_prof_filter.empty() || g_prof_filter == name)) return false;
    return g_prof_period <= 1 || (g_prof_seen++ % g_prof_period) == 0;   // (every n-th matching launch)
}
extern "C" int mgr_profile_sample_every(int n) {
    g_prof_period = n < 1 ? 1 : n;
    g_prof_seen = 0;
    return MGR_OK;
}

extern "C" int mgr_profile_enable(int on) {
    g_mgr_prof_on = on ? 1 : 0;
    return MGR_OK;
}

extern "C" int mgr_profile_filter(const char* kernel_name) {
    g_prof_filter = kernel_name ? kernel_name : "";
    return MGR_OK;
}

extern "C" int mgr_profile_report(char* buf, size_t len, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MGR_HIP(hipStreamSynchronize(stream));
    std::map<std::string, std::pair<long, double>> agg;
    std::vector<std::string> order;
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto it = agg.find(r.name);
            if (it == agg.end()) { order.push_back(r.name); agg[r.name] = {1, ms}; }
            else { it->second.first += 1; it->second.second += ms; }
        }
        if (r.a) g_prof_pool.push_back(r.a);
        if (r.b) g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    std::string out;
    for (auto& n : order) {
        char line[256];
        snprintf(line, sizeof(line), "%s %ld %.6f\n", n.c_str(), agg[n].first, agg[n].second);
        out += line;
    }
    if (buf && len) {
        size_t k = out.size() < len - 1 ? out.size() : len - 1;
        memcpy(buf, out.data(), k);
        buf[k] = 0;
    }
    return MGR_OK;
}

extern "C" int mgr_version(void) { return MGR_VERSION; }
// instrumentation compiled into the library (include/manus_hip.h: 0 for the product build); the backward's half lives in
// its translation unit, the macros being per-source -D flags of tools/instr/build_variant.sh
int mgr_bwd_variant_bits(void);
extern "C" int mgr_build_variant(void) {
    int bits = mgr_bwd_variant_bits();
#ifdef FWD_KO_DEEP
    bits |= 1;
#endif
#if defined(MGR_STATS) || defined(MGR_TIMELINE) || defined(FWD_PROF) || defined(BIN_PROF) || defined(DBS_PROF)
    bits |= 2;
#endif
#if defined(FWD_PF1) || defined(FWD_LDS_PIPE1)
    bits |= 4;
#endif
    if (MGR_BIN_BLOCK != 1024 || MGR_GREC_BYTES != 64 || MGR_DB_BUCKETS != 1024 || MGR_DB_ITEM != 768 || MGR_DB_RANK_MAX != 2048 ||
        MGR_DB_RANK_LARGE != 3072)
        bits |= 4;      // (round 6's A/B switches of the instance sort)
    return bits;
}
extern "C" const char* mgr_last_error(void) { return g_mgr_err; }

extern "C" int mgr_raster_record_bytes(void) { return (int)sizeof(MgrGRec); }
extern "C" size_t mgr_raster_workspace_bytes(int V, int N, int W, int H, int64_t cap) {
    return mgr_layout(V, N, W, H, cap).total;
}

// ---------------------------------------------------------------------------
// block-wide exclusive scan of one uint per thread (blockDim.x = 1024 or 256)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t val, uint32_t* s_wave /*>=17*/,
                                                    uint32_t& block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t incl = mgr_wave_incl_scan_u32(val);   // (DPP: no ds_bpermute round trips)
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < nw; ++w) {
            uint32_t t = s_wave[w];
            s_wave[w] = run;
            run += t;
        }
        s_wave[nw] = run;
    }
    __syncthreads();
    block_total = s_wave[nw];
    uint32_t r = s_wave[wave] + incl - val;
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------
// K1: per-Gaussian projection, EWA conic, radius, tile rectangle; per-tile
// counts (LDS-aggregated); pair-slot offsets (block scan + one atomic per block)
// ---------------------------------------------------------------------------
// Workgroup sizes of the per-instance kernels (measured on the bench workload: k_inst_fwd 0.245 / 0.173 / 0.189 ms
// at 1024 / 512 / 256 threads, k_emit 0.100 / 0.105 / 0.175 ms: the LDS tile histogram is flushed once per workgroup).
// (late in round 4, same kernel with the depth-cut lookups: 0.198 / 0.192 / 0.172 / 0.215 / 0.179 ms at 384 / 448 / 512 / 640 / 768.)
#ifndef MGR_FWD_GRID
// Workgroups of k_blend_fwd.  Workgroup w starts with queue position w (deepest tiles first) and draws tickets behind the grid;
// ~1280 of them are resident at a time (five per CU), the others are handed to the CUs by the hardware dispatcher as slots
// free up -- in blockIdx order, i.e. down the depth-ordered queue, with no ticket round trip.  Round 5 sweep on the no-hints
// step (k_blend_fwd, ms): 1536 0.2615 | 2048 0.2592 | 3072 0.2517 | 4096 0.2480 | 6144 0.2445 | 8192 0.2447; the 500 k composite
// with 7 views 0.333 -> 0.280; one and four views unchanged.  (Round 4 had looked at 1024 .. 2048 only.)
#define MGR_FWD_GRID (256 * 24)
#endif
#ifndef PRE_THREADS
#define PRE_THREADS 512
#endif
#define EMIT_THREADS 1024

// Depth bucket of an instance: uniform over the depth range [zmin, zmax] of the view's visible instances (round 6; up to round 5:
// 1024 buckets per octave of z above the cull plane -- a hand at 1.2 m then fills ~200 of the 8192 buckets with ~1000 keys each,
// and a sort item could not be smaller than a bucket).  Monotone in z (a float subtraction and a multiplication by a positive
// constant are), evaluated by the same function wherever a bucket is needed.
// The range: every workgroup of the per-instance kernel leaves (largest complement of its visible depths' bits, largest bits) in
// its own slot of db_zrange -- a plain store behind its last barrier: no atomics, nothing to wait for (sending them to sixteen
// shared words by atomicMax in front of a barrier cost that kernel 9 - 21 us; the previous forward's range, tried next, misfits
// whenever the views change: end buckets of thousands of keys) -- and the bucket kernels reduce the view's slots (a few KB).
struct DbRange { float zmin, scale; };
__device__ __forceinline__ DbRange db_range_reduce(const uint32_t* __restrict__ slots /* this view's */, int nslots, uint32_t* s_zr /* LDS, 2 words, zeroed */) {
    uint32_t lo_c = 0u, hi = 0u;
    for (int k = threadIdx.x; k < nslots; k += blockDim.x) { lo_c = max(lo_c, slots[2 * k]); hi = max(hi, slots[2 * k + 1]); }
    lo_c = mgr_wave_max_u32(lo_c); hi = mgr_wave_max_u32(hi);
    if ((threadIdx.x & 63) == 0 && hi) { atomicMax(&s_zr[0], lo_c); atomicMax(&s_zr[1], hi); }
    __syncthreads();
    lo_c = s_zr[0]; hi = s_zr[1];
    DbRange r;
    r.zmin = __uint_as_float(~lo_c);
    const float ext = __uint_as_float(hi) - r.zmin;
    r.scale = (hi != 0u && ext > 0.0f) ? (float)(MGR_DB_BUCKETS - 1) / ext : 0.0f;
    return r;
}
__device__ __forceinline__ uint32_t db_bucket(float z, const DbRange r) {
    const float f = (z - r.zmin) * r.scale;
    return f > 0.0f ? min((uint32_t)f, (uint32_t)MGR_DB_BUCKETS - 1u) : 0u;
}
// an instance takes part when it has at least one non-null tile
__device__ __forceinline__ bool db_takes_part(int radius, ushort4 rc, unsigned long long am) {
    const uint32_t tiles = (uint32_t)((rc.z - rc.x) * (rc.w - rc.y));
    return radius > 0 && tiles > 0u && (tiles > 64u || am != 0ull);
}

// Shared tail of the per-instance forward kernels: per-tile histogram over the NON-NULL tiles of
// the rectangle (exact culling, mgr_box_dead; the mask is stored so that k_emit makes the
// identical decision; rectangles of more than 64 tiles are not culled), pair-slot offsets
// (contiguous per Gaussian, block base from one atomic), and the per-instance state records.
__device__ __forceinline__ void pre_tail(int N, int gx, int gy, int v, int i, const ProjOut& po, float op_i,
                                         const float col[3], uint32_t* s_scan, uint32_t* s_hist, int lds_hist,
                                         MgrGRec* __restrict__ grec, float* __restrict__ depth,
                                         ushort4* __restrict__ rect, unsigned long long* __restrict__ alive,
                                         uint32_t* __restrict__ pair_off, uint32_t* __restrict__ tile_count,
                                         int32_t* __restrict__ radii, MgrHeader* hdr, uint32_t* __restrict__ zr_slot,
                                         const uint32_t* __restrict__ zcut_v = nullptr) {
    const int tid = threadIdx.x, T = gx * gy;
    const int radius = po.radius, x0 = po.x0, y0 = po.y0, x1 = po.x1, y1 = po.y1;
    const uint32_t tiles = (uint32_t)((x1 - x0) * (y1 - y0));
    {   // depth range of this workgroup's visible instances (db_range_reduce): wave maxima -> s_scan[22 / 23] (zeroed by the kernel
        // before its first barrier); stored to the workgroup's slot at the very end
        const uint32_t zb = (radius > 0 && tiles > 0u) ? __float_as_uint(po.zv) : 0u;
        const uint32_t wmax = mgr_wave_max_u32(zb), wminc = mgr_wave_max_u32(zb ? ~zb : 0u);
        if ((tid & 63) == 0 && wmax) { atomicMax(&s_scan[23], wmax); atomicMax(&s_scan[22], wminc); }
    }
    unsigned long long amask = ~0ull;
    if (radius > 0 && !(zcut_v && tiles <= 64)) {
        const bool small = tiles <= 64;
        const MgrCull cull = mgr_cull_init(po.px, po.py, po.ca, po.cb, po.cc, mgr_qmax(op_i));
        if (small) amask = 0ull;
        int k = 0;
        for (int y = y0; y < y1; ++y) {
            float dy_lo, dy_hi, dxo;
            mgr_cull_row(cull, 16.0f * y, 16.0f * y + 15.0f, dy_lo, dy_hi, dxo);
            for (int x = x0; x < x1; ++x, ++k) {
                if (small) {
                    if (mgr_cull_dead(cull, dy_lo, dy_hi, dxo, 16.0f * x, 16.0f * x + 15.0f)) continue;
                    amask |= 1ull << k;
                }
                if (lds_hist) atomicAdd(&s_hist[y * gx + x], 1u);
                else atomicAdd(&tile_count[(size_t)v * T + y * gx + x], 1u);
            }
        }
    } else if (radius > 0) {
        // Depth cut (zcut_v: this view's row of tile_zcut).  A tile whose every pixel saturated in front of depth zc in the
        // previous forward of this view leaves out the instances behind zc -- they sit behind the stop of every pixel
        // unless the scene changed, which the blend detects (MGR_OVF_CUT) and the caller answers by a forward without the
        // cut.  Stored value = ~(float bits of zc), 0 = no cut; depths are positive, so their bits order like the floats.
        // The lookup of the next tile is issued before the current one is tested, so that the loop does not wait a memory
        // round trip per tile.  (Measured: lookups behind the test of their own tile +16 us on k_inst_fwd; two passes -- the
        // null-tile test, then the survivors' lookups four at a time -- +8 us, 15 % more VALU instructions.  Rectangles of
        // more than 64 tiles have no mask and are never cut.)
        const MgrCull cull = mgr_cull_init(po.px, po.py, po.ca, po.cb, po.cc, mgr_qmax(op_i));
        const uint32_t zbits = __float_as_uint(po.zv);
        amask = 0ull;
        int k = 0;
        uint32_t zc = tiles ? zcut_v[y0 * gx + x0] : 0u;      // (the lookup of tile k + 1 is in flight while tile k is tested)
        for (int y = y0; y < y1; ++y) {
            float dy_lo, dy_hi, dxo;
            mgr_cull_row(cull, 16.0f * y, 16.0f * y + 15.0f, dy_lo, dy_hi, dxo);
            for (int x = x0; x < x1; ++x, ++k) {
                const bool wrap = x + 1 == x1;
                const int nx = wrap ? x0 : x + 1, ny = wrap ? y + 1 : y;
                const uint32_t zn = zcut_v[min(ny, y1 - 1) * gx + nx];
                const bool dead = mgr_cull_dead(cull, dy_lo, dy_hi, dxo, 16.0f * x, 16.0f * x + 15.0f);
                if (!dead && zbits <= ~zc) {
                    amask |= 1ull << k;
                    if (lds_hist) atomicAdd(&s_hist[y * gx + x], 1u);
                    else atomicAdd(&tile_count[(size_t)v * T + y * gx + x], 1u);
                }
                zc = zn;
            }
        }
    }
    uint32_t block_total;
    const uint32_t local = block_excl_scan(tiles, s_scan, block_total);
    if (tid == 0) s_scan[20] = block_total ? atomicAdd(&hdr->acc_pairs, block_total) : 0u;
    __syncthreads();
    const uint32_t off = s_scan[20] + local;
    if (i < N) {
        const size_t vi = (size_t)v * N + i;
        MgrGRec r;
        r.x = po.px; r.y = po.py; r.ca = po.ca; r.cb = po.cb; r.cc = po.cc;
        r.op = op_i;
        r.r = col[0]; r.g = col[1]; r.b = col[2];
        r.rect_w = x1 - x0;
        r.slot_base = (int32_t)off - y0 * (x1 - x0) - x0;
        r.pad = 0;
        grec[vi] = r;
        depth[vi] = po.zv;
        rect[vi] = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
        alive[vi] = amask;
        pair_off[vi] = off;
        radii[vi] = radius;
    }
    if (lds_hist) {
        __syncthreads();
        for (int k = tid; k < T; k += PRE_THREADS) {
            const uint32_t c = s_hist[k];
            if (c) atomicAdd(&tile_count[(size_t)v * T + k], c);
        }
    }
    if (tid == 0) { zr_slot[0] = s_scan[22]; zr_slot[1] = s_scan[23]; }      // (behind the scan's barriers: every wave's maxima are in)
}

__global__ __launch_bounds__(PRE_THREADS) void k_preprocess(
    int N, int W, int H, int gx, int gy, const float* __restrict__ cams,
    const float* __restrict__ means3D, int64_t s_means, const float* __restrict__ cov3D,
    int64_t s_cov, const float* __restrict__ colors, int64_t s_col,
    const float* __restrict__ opacity, int64_t s_op, MgrGRec* __restrict__ grec,
    float* __restrict__ depth, ushort4* __restrict__ rect, unsigned long long* __restrict__ alive,
    uint32_t* __restrict__ pair_off, uint32_t* __restrict__ tile_count, int32_t* __restrict__ radii,
    MgrHeader* hdr, int lds_hist, uint32_t* __restrict__ db_zrange) {
    extern __shared__ uint32_t s_mem[];
    uint32_t* s_scan = s_mem;       // 32 words
    uint32_t* s_hist = s_mem + 32;  // gx*gy words when lds_hist
    const int v = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * PRE_THREADS + tid;
    const int T = gx * gy;
    if (lds_hist) {
        for (int k = tid; k < T; k += PRE_THREADS) s_hist[k] = 0;
    }
    if (tid < 2) s_scan[22 + tid] = 0u;
    __syncthreads();

    MgrCam cam;
    mgr_load_cam(cams, v, cam);

    ProjOut po;
    po.radius = 0; po.x0 = po.y0 = po.x1 = po.y1 = 0;
    po.px = po.py = po.ca = po.cb = po.cc = po.zv = 0.f;
    if (i < N) {
        const float* mp = means3D + (size_t)v * s_means + (size_t)i * 3;
        const float p[3] = {mp[0], mp[1], mp[2]};
        const float* cp = cov3D + (size_t)v * s_cov + (size_t)i * 6;
        const float c6[6] = {cp[0], cp[1], cp[2], cp[3], cp[4], cp[5]};
        project_gaussian(cam, W, H, gx, gy, p, c6, po);
    }
    const float op_i = (i < N) ? opacity[(size_t)v * s_op + i] : 0.0f;
    float col[3] = {0.f, 0.f, 0.f};
    if (i < N) {
        const float* cp = colors + (size_t)v * s_col + (size_t)i * 3;
        col[0] = cp[0]; col[1] = cp[1]; col[2] = cp[2];
    }
    pre_tail(N, gx, gy, v, i, po, op_i, col, s_scan, s_hist, lds_hist, grec, depth, rect, alive, pair_off, tile_count,
             radii, hdr, db_zrange + 2 * ((size_t)v * gridDim.x + blockIdx.x));
}

// ---------------------------------------------------------------------------
// Fused per-instance forward: canonical parameters -> (LBS of mean and covariance) -> SH colour
// (view direction pulled back through the blended transform) -> sigmoid opacity -> projection
// and binning state, without writing posed means / covariances / transforms / colours to HBM.
// Same math as k_lbs_fwd + k_sh_fwd + k_preprocess (instance_math.h).
// skin_w == nullptr: static object (identity transform, direction = xyz - camera).
// Rows >= n_art are static as well (the object half of a hand+object composite, src/modules/composite.py:50-59:
// tf = identity, so inv(tf) * cam = cam exactly and the colour equals the static route's).
// MIXED = false: all Gaussians articulated (or none): the route is a kernel-argument test, uniform for the launch.
// ---------------------------------------------------------------------------
template <bool MIXED, bool SH_HALF>
__global__ __launch_bounds__(PRE_THREADS) void k_inst_fwd(
    int N, int B, int n_art, int W, int H, int gx, int gy, const float* __restrict__ cams,
    const float* __restrict__ xyz, const float* __restrict__ log_scale, const float* __restrict__ rot,
    const float* __restrict__ op_logit, const float* __restrict__ f_dc, const float* __restrict__ f_rest,
    const float* __restrict__ skin_w, const float* __restrict__ transforms, MgrGRec* __restrict__ grec,
    float* __restrict__ depth, ushort4* __restrict__ rect, unsigned long long* __restrict__ alive,
    uint32_t* __restrict__ pair_off, uint32_t* __restrict__ tile_count, int32_t* __restrict__ radii,
    MgrHeader* hdr, int lds_hist, int V, const uint32_t* __restrict__ tile_zcut, uint32_t* __restrict__ db_zrange) {
    extern __shared__ uint32_t s_mem[];
    uint32_t* s_scan = s_mem;
    uint32_t* s_hist = s_mem + 32;
    // XCD-aware block order.  The V views of a range of 512 Gaussians read the same canonical rows (320 B per
    // Gaussian, the SH coefficients most of it): workgroup b runs on XCD b % 8 (observed dispatch order, used for
    // speed only), so the ranges are dealt out to the XCDs (range r -> XCD r % 8) and the V workgroups of a range
    // follow each other on that XCD -- views 2..V find the rows in its L2 (a range is 169 KB, an XCD holds ~16 ranges
    // in flight, its L2 is 4 MB).  With (range, view) = (blockIdx.x, blockIdx.y) the next view of a range came 586
    // workgroups later on some other XCD and every view fetched its rows from memory again.
    const int tid = threadIdx.x;
    const int nrange = (N + PRE_THREADS - 1) / PRE_THREADS;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int range = (k / V) * 8 + xcd, v = k % V;
    if (range >= nrange) return;   // (the grid is padded to 8 * ceil(nrange / 8) * V workgroups)
    const int i = range * PRE_THREADS + tid;
    const int T = gx * gy;
    if (lds_hist) {
        for (int k2 = tid; k2 < T; k2 += PRE_THREADS) s_hist[k2] = 0;
    }
    if (tid < 2) s_scan[22 + tid] = 0u;
    __syncthreads();
    MgrCam cam;
    mgr_load_cam(cams, v, cam);
    ProjOut po;
    po.radius = 0; po.x0 = po.y0 = po.x1 = po.y1 = 0;
    po.px = po.py = po.ca = po.cb = po.cc = po.zv = 0.f;
    float col[3] = {0.f, 0.f, 0.f}, op_i = 0.f;
    if (i < N) {
        GaussCano g;
        cano_load(xyz, log_scale, rot, i, g);
        float tf[12], p[3], c6[6];
        const bool art = skin_w != nullptr && (!MIXED || i < n_art);
        blend_tf(art ? skin_w + (size_t)i * B : nullptr, art ? transforms + (size_t)v * B * 16 : nullptr, B, tf);
        lbs_apply(tf, g, p, c6);
        project_gaussian(cam, W, H, gx, gy, p, c6, po);
        op_i = 1.0f / (1.0f + expf(-op_logit[i]));
        if (po.radius > 0) {  // colour is only consumed by the blend
            ShDir D;
            if (art) sh_dir_xyz<true>(g.x, g.y, g.z, tf, cam.campos, D);
            else sh_dir_xyz<false>(g.x, g.y, g.z, tf, cam.campos, D);
            float Y[16], c[48], rgb[3];
            sh_basis(D.d[0] / D.n, D.d[1] / D.n, D.d[2] / D.n, Y);
            const mgr_f3u d = *(const mgr_f3u*)(f_dc + (size_t)i * 3);
            c[0] = d.x; c[1] = d.y; c[2] = d.z;
            if (SH_HALF) {   // fp16 storage: 48 halves per row (45 used), six aligned 16-byte loads
                const uint4* hr = (const uint4*)((const __half*)f_rest + (size_t)i * MGR_SH_HALF_ROW);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const uint4 t = hr[q];
                    const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 8 * q + 2 * e;
                        const __half2 h2 = *reinterpret_cast<const __half2*>(&w4[e]);
                        if (k < 45) c[3 + k] = __low2float(h2);
                        if (k + 1 < 45) c[4 + k] = __high2float(h2);
                    }
                }
            } else {   // 45 coefficients in 12 loads
                const float* fr = f_rest + (size_t)i * 45;
#pragma unroll
                for (int q = 0; q < 11; ++q) {
                    const mgr_f4u t = *(const mgr_f4u*)(fr + 4 * q);
                    c[3 + 4 * q] = t.x; c[4 + 4 * q] = t.y; c[5 + 4 * q] = t.z; c[6 + 4 * q] = t.w;
                }
                c[47] = fr[44];
            }
            sh_rgb(c, Y, rgb);
            col[0] = fmaxf(rgb[0] + 0.5f, 0.f);
            col[1] = fmaxf(rgb[1] + 0.5f, 0.f);
            col[2] = fmaxf(rgb[2] + 0.5f, 0.f);
        }
    }
    pre_tail(N, gx, gy, v, i, po, op_i, col, s_scan, s_hist, lds_hist, grec, depth, rect, alive, pair_off, tile_count,
             radii, hdr, db_zrange + 2 * ((size_t)v * nrange + range), tile_zcut ? tile_zcut + (size_t)v * T : nullptr);
}

// ---------------------------------------------------------------------------
// K2: exclusive scan of the V*T tile counts -> tile_start; queue of non-empty
// tiles ordered by descending size class (largest first = LPT scheduling)
// ---------------------------------------------------------------------------
// Queue key.  The forward blend takes the tiles largest first (LPT), and what a tile costs is the list depth its pixels
// consume, not the list's length: the per-tile timeline (-DMGR_TIMELINE=2) showed lists of 4 000-8 000 entries that saturate
// after 300-900 (30-90 us) at the head of the queue while never-saturating lists of 3 000-4 000 entries (120 us) started at
// 0.25 of a 0.39 ms kernel and ended it.  The depth a tile consumed in the PREVIOUS forward on this workspace (tile_done,
// still in place when the scan runs) predicts it well whenever consecutive steps render similar views -- a fixed camera
// rig, a slowly changing model.  It is a scheduling hint only: a wrong hint costs balance, never correctness; a fresh
// workspace (all zero) or a tile that was empty falls back to the list length.
// (Only with the depth-ordered binning: the per-tile sort route cuts its queue at list LENGTH classes.)
__device__ __forceinline__ uint32_t mgr_queue_key(uint32_t count, uint32_t prev_done, int use_hint) {
    return (use_hint && count && prev_done) ? min(count, prev_done) : count;
}

#define DB_PER 8   // instances per thread of the bucket scatter (one LDS histogram flush per 8192 instances)
// The depth-bucket kernels ride in other launches where they can: the count (8192 instances of one view per workgroup)
// behind the workgroups of tile-scan phase A, the bucket scan (one workgroup per view) behind those of phase B -- each
// depends on the launch before only, and as launches of their own they cost 20 + 15 us for a few microseconds of work.
// (Counting the buckets with global atomics from the per-instance kernel instead was measured: the hand's instances fall
// into ~200 buckets per view and k_inst_fwd went from 0.17 to 0.36 ms.)
__device__ __forceinline__ void dbin_count_block(int bx, int v, int N, const int32_t* __restrict__ radii,
                                                 const float* __restrict__ depth, const ushort4* __restrict__ rect,
                                                 const unsigned long long* __restrict__ alive, uint32_t* __restrict__ db_count,
                                                 uint32_t* s_hist /*MGR_DB_BUCKETS*/, const uint32_t* __restrict__ db_zrange, int nslots) {
    const int tid = threadIdx.x;
    __shared__ uint32_t s_zr[2];
    if (tid < 2) s_zr[tid] = 0u;
    for (int k = tid; k < MGR_DB_BUCKETS; k += 1024) s_hist[k] = 0;
    // (the instances' loads are issued in front of the range's barriers: one dependent round trip less)
    float z[DB_PER];
    uint32_t on = 0;
#pragma unroll
    for (int r = 0; r < DB_PER; ++r) {
        const int i = (bx * DB_PER + r) * 1024 + tid;
        z[r] = 0.f;
        if (i < N) {
            const size_t vi = (size_t)v * N + i;
            z[r] = depth[vi];
            if (db_takes_part(radii[vi], rect[vi], alive[vi])) on |= 1u << r;
        }
    }
    __syncthreads();
    const DbRange zr = db_range_reduce(db_zrange + 2 * (size_t)v * nslots, nslots, s_zr);
#pragma unroll
    for (int r = 0; r < DB_PER; ++r)
        if ((on >> r) & 1u) atomicAdd(&s_hist[db_bucket(z[r], zr)], 1u);
    __syncthreads();
    for (int k = tid; k < MGR_DB_BUCKETS; k += 1024) {
        const uint32_t c = s_hist[k];
        if (c) atomicAdd(&db_count[(size_t)v * MGR_DB_BUCKETS + k], c);
    }
}

// One workgroup per view (extra workgroups of the k_tile_scan_b launch): bucket offsets inside the view's segment, the
// number of participating instances (the counters are consumed: left zero for the next forward), and the bounding box of the
// view's non-empty tiles (from the boxes phase A left per block of tiles).
__device__ __forceinline__ void dbin_scan_block(int v, int nbT, const uint4* __restrict__ blk_box,
                                                uint32_t* __restrict__ db_count, uint32_t* __restrict__ db_cursor,
                                                uint32_t* __restrict__ db_start, uint32_t* __restrict__ db_nvis,
                                                ushort4* __restrict__ db_bbox, uint32_t* s_scan /*32*/, uint32_t* s_box /*4*/,
                                                MgrHeader* hdr, uint32_t* __restrict__ db_item, int items_per_view, uint32_t* s_items /* LDS, 2 x items_per_view words, or nullptr */) {
    const int tid = threadIdx.x;
    if (tid == 0) { s_box[0] = 0xFFFFu; s_box[1] = 0xFFFFu; s_box[2] = 0u; s_box[3] = 0u; }
    if (s_items)
        for (int k = tid; k < 2 * items_per_view; k += 1024) s_items[k] = 0u;      // (published by the scan's barrier below)
    constexpr int PER = MGR_DB_BUCKETS / 1024;
    uint32_t c[PER], sum = 0;
    uint32_t* cnt = db_count + (size_t)v * MGR_DB_BUCKETS + tid * PER;
#pragma unroll
    for (int k = 0; k < PER; ++k) { c[k] = cnt[k]; sum += c[k]; cnt[k] = 0; }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, s_scan, total);   // (contains the barrier that publishes s_box)
    // sort items: item i of the view = the buckets whose first key lies in [i, i + 1) * MGR_DB_ITEM of the bucketed order; every
    // non-empty bucket reports itself to its item -- in LDS (s_items: first bucket as a complement, end of the last: both kept by
    // atomicMax on zeros); the table is then written out whole, empty items as zeros (global atomics from here measured 40 us)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        db_start[(size_t)v * (MGR_DB_BUCKETS + 1) + tid * PER + k] = run;
        db_cursor[(size_t)v * MGR_DB_BUCKETS + tid * PER + k] = 0;
        if (c[k] && s_items) {
            const uint32_t it = min(run / (uint32_t)MGR_DB_ITEM, (uint32_t)items_per_view - 1u), b = (uint32_t)(tid * PER + k);
            atomicMax(&s_items[2 * it], ~b);
            atomicMax(&s_items[2 * it + 1], b + 1u);
        }
        run += c[k];
    }
    uint32_t x0 = 0xFFFFu, y0 = 0xFFFFu, x1 = 0u, y1 = 0u;
    if (tid < nbT) {
        const uint4 bb = blk_box[(size_t)v * nbT + tid];
        x0 = bb.x; y0 = bb.y; x1 = bb.z; y1 = bb.w;
    }
    if (x1 > 0u) { atomicMin(&s_box[0], x0); atomicMin(&s_box[1], y0); atomicMax(&s_box[2], x1); atomicMax(&s_box[3], y1); }
    __syncthreads();
    if (s_items)
        for (int k = tid; k < 2 * items_per_view; k += 1024) db_item[(size_t)v * items_per_view * 2 + k] = s_items[k];
    if (tid == 0) {
        db_start[(size_t)v * (MGR_DB_BUCKETS + 1) + MGR_DB_BUCKETS] = total;
        db_nvis[v] = total;
        const bool any = s_box[2] > 0u;   // box = (x0, y0, width, height); empty view: 0 x 0
        db_bbox[v] = any ? make_ushort4((unsigned short)s_box[0], (unsigned short)s_box[1], (unsigned short)(s_box[2] - s_box[0]),
                                        (unsigned short)(s_box[3] - s_box[1]))
                         : make_ushort4(0, 0, 0, 0);
        const uint32_t tb = any ? (s_box[2] - s_box[0]) * (s_box[3] - s_box[1]) : 0u;   // which LDS tier of the bin kernels the view needs
        if (tb > 2048u) atomicOr(&hdr->tiers, 1u);
        else if (tb > 1536u) atomicOr(&hdr->tiers, 2u);
    }
}

// 8192 instances of one view per workgroup (extra workgroups of the k_tile_scan_b launch): keys into their depth buckets.
__device__ __forceinline__ void dbin_scatter_block(int bx, int v, int N, const int32_t* __restrict__ radii,
                                                   const float* __restrict__ depth, const ushort4* __restrict__ rect,
                                                   const unsigned long long* __restrict__ alive,
                                                   const uint32_t* __restrict__ db_start, uint32_t* __restrict__ db_cursor,
                                                   unsigned long long* __restrict__ db_keys, uint32_t* s_hist /*MGR_DB_BUCKETS*/,
                                                   const uint32_t* __restrict__ db_zrange, int nslots) {
    const int tid = threadIdx.x;
    __shared__ uint32_t s_zr[2];
    if (tid < 2) s_zr[tid] = 0u;
    for (int k = tid; k < MGR_DB_BUCKETS; k += 1024) s_hist[k] = 0;
    float z[DB_PER];
    uint32_t on = 0;
#pragma unroll
    for (int r = 0; r < DB_PER; ++r) {      // (in front of the range's barriers, like dbin_count_block)
        const int i = (bx * DB_PER + r) * 1024 + tid;
        z[r] = 0.f;
        if (i < N) {
            const size_t vi = (size_t)v * N + i;
            z[r] = depth[vi];
            if (db_takes_part(radii[vi], rect[vi], alive[vi])) on |= 1u << r;
        }
    }
    __syncthreads();
    const DbRange zr = db_range_reduce(db_zrange + 2 * (size_t)v * nslots, nslots, s_zr);
#pragma unroll
    for (int r = 0; r < DB_PER; ++r)
        if ((on >> r) & 1u) atomicAdd(&s_hist[db_bucket(z[r], zr)], 1u);
    __syncthreads();
    for (int k = tid; k < MGR_DB_BUCKETS; k += 1024) {
        const uint32_t c = s_hist[k];
        if (c) s_hist[k] = db_start[(size_t)v * (MGR_DB_BUCKETS + 1) + k] + atomicAdd(&db_cursor[(size_t)v * MGR_DB_BUCKETS + k], c);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DB_PER; ++r) {
        if ((on >> r) & 1u) {
            const int i = (bx * DB_PER + r) * 1024 + tid;
            const uint32_t pos = atomicAdd(&s_hist[db_bucket(z[r], zr)], 1u);
            db_keys[(size_t)v * N + pos] = ((unsigned long long)__float_as_uint(z[r]) << 32) | (unsigned)i;
        }
    }
}

__global__ __launch_bounds__(1024) void k_dbin_scatter(int N, const int32_t* __restrict__ radii,
                                                       const float* __restrict__ depth, const ushort4* __restrict__ rect,
                                                       const unsigned long long* __restrict__ alive,
                                                       const uint32_t* __restrict__ db_start, uint32_t* __restrict__ db_cursor,
                                                       unsigned long long* __restrict__ db_keys, const uint32_t* __restrict__ db_zrange, int nslots) {
    __shared__ uint32_t s_hist[MGR_DB_BUCKETS];
    dbin_scatter_block((int)blockIdx.x, (int)blockIdx.y, N, radii, depth, rect, alive, db_start, db_cursor, db_keys, s_hist, db_zrange, nslots);
}

struct DbinArgs {   // depth-bucket side of the two tile-scan launches (ordered binning); db_count == nullptr: none
    int N, n_bx;
    const int32_t* radii;
    const float* depth;
    const ushort4* rect;
    const unsigned long long* alive;
    uint32_t *db_count, *db_cursor, *db_start, *db_nvis;
    ushort4* db_bbox;
    unsigned long long* db_keys;
    uint32_t *db_zrange, *db_item;
    int items_per_view, zr_slots;      // (zr_slots: workgroups of the per-instance kernel per view = slots of db_zrange per view)
};

#define MGR_HOLE 0xFFFFFFFEu   // a position of the view-interleaved queue its view has no tile for
#define DBR_MAX_ITEMS 4096    // sort items per view the bucket scan can table in LDS (1 M instances per view; beyond: the radix launch sorts everything)
#define MGR_NCLS 34

// Phase A: one block = up to 1024 tiles of ONE view (grid = V x ceil(T / 1024), view-major like the tile index, so the
// block sums are in tile order).  Per block: the sums of pairs and checkpoints and the histogram of its tiles over the size
// classes -- written, not accumulated: phase B re-derives every base from these few numbers, so the queues need neither
// global atomics nor counters that somebody has to clear.
__global__ __launch_bounds__(1024) void k_tile_scan_a(int T, int nbT, const uint32_t* __restrict__ tile_count,
                                                      const uint32_t* __restrict__ tile_done, int use_hint,
                                                      uint2* __restrict__ part, uint32_t* __restrict__ blk_cls,
                                                      uint4* __restrict__ blk_box, int gx, int n_scan_blocks, DbinArgs db,
                                                      MgrHeader* hdr) {
    __shared__ uint32_t s_dbh[MGR_DB_BUCKETS];
    if ((int)blockIdx.x >= n_scan_blocks) {   // the workgroups behind the scan's: depth buckets of 8192 instances of one view
        const int b2 = (int)blockIdx.x - n_scan_blocks;
        dbin_count_block(b2 % db.n_bx, b2 / db.n_bx, db.N, db.radii, db.depth, db.rect, db.alive, db.db_count, s_dbh, db.db_zrange, db.zr_slots);
        return;
    }
    __shared__ uint32_t s_scan[32];
    __shared__ uint32_t s_cls[MGR_NCLS];
    __shared__ uint32_t s_bb[4];
    const int tid = threadIdx.x, v = blockIdx.x / nbT, t = (blockIdx.x % nbT) * 1024 + tid;
    if (tid == 0) { s_bb[0] = 0xFFFFu; s_bb[1] = 0xFFFFu; s_bb[2] = 0u; s_bb[3] = 0u; }
    if (blockIdx.x == 0 && tid == 0) { hdr->tiers = 0u; hdr->sort_big = 0u; hdr->sort_huge = 0u; hdr->sort_large = 0u; hdr->sort_near_large = 0u; hdr->fwd_seq += 1u; hdr->rep_why = 0u; }   // (phase B's depth-bucket workgroups / the sort set them)
    const bool valid = t < T;
    const size_t k = (size_t)v * T + t;
    if (tid < MGR_NCLS) s_cls[tid] = 0;
    __syncthreads();
    const uint32_t c = valid ? tile_count[k] : 0u;
    const uint32_t key = mgr_queue_key(c, valid ? tile_done[k] : 0u, use_hint);
    {   // (class 0 = empty tiles, 85 % of a capture-like frame: counted per wave by a ballot -- a thousand LDS atomics on one
        // address are served one after the other)
        const int cls = key ? 32 - __clz(key) : 0;
        const unsigned long long m0 = __ballot(valid && cls == 0);
        if (valid && cls != 0) atomicAdd(&s_cls[cls], 1u);
        if ((tid & 63) == 0 && m0) atomicAdd(&s_cls[0], (uint32_t)__popcll(m0));
    }
    if (c) {   // bounding box of the block's non-empty tiles (one wave-level step, then four LDS atomics per wave)
        const uint32_t y = (uint32_t)t / (uint32_t)gx, x = (uint32_t)t - y * (uint32_t)gx;
        atomicMin(&s_bb[0], x); atomicMin(&s_bb[1], y); atomicMax(&s_bb[2], x + 1u); atomicMax(&s_bb[3], y + 1u);
    }
    uint32_t total, ctotal;
    (void)block_excl_scan(c, s_scan, total);
    (void)block_excl_scan(c ? (c - 1) / MGR_CHUNK : 0u, s_scan, ctotal);
    if (tid == 0) part[blockIdx.x] = make_uint2(total, ctotal);
    if (tid < MGR_NCLS) blk_cls[(size_t)blockIdx.x * MGR_NCLS + tid] = s_cls[tid];
    if (tid == 0) blk_box[blockIdx.x] = make_uint4(s_bb[0], s_bb[1], s_bb[2], s_bb[3]);   // (behind the scans' barriers)
}

// Phase B: every block re-derives its bases from the block sums and block histograms, writes tile_start / chunk_start,
// and places its tiles in two queues:
//  * tile_queue -- all tiles of all views by descending size class, empty tiles last (the per-tile sort route cuts it
//    at class boundaries; the forward blend background-fills its empty tail);
//  * tile_qrec -- the forward blend's queue, INTERLEAVED BY VIEW: position p holds the (p / V)-th deepest tile of view
//    p % V as a self-contained record (tile, list offset, list length, first checkpoint), or a hole when that view has
//    fewer tiles.  MgrQueue's counter c hands out the positions c, c + 16, ... and a workgroup's home counter is its index
//    mod 16, so with 8 views the workgroups of XCD x -- every eighth -- blend the tiles of view x while it has any: each L2
//    holds ONE view's Gaussian records (measured: 27 % of the blend's gathers that miss the L1 hit the L2 with views
//    mixed over the XCDs; k_blend_fwd 0.367 -> 0.325 ms with this order).  A scheduling matter only.
// Order inside a (view, class) is by block, then arbitrary inside a block.
__global__ __launch_bounds__(1024) void k_tile_scan_b(int V, int T, int nbT, uint32_t* __restrict__ tile_count,
                                                      const uint2* __restrict__ part, const uint32_t* __restrict__ blk_cls,
                                                      uint32_t* __restrict__ tile_start,
                                                      uint32_t* __restrict__ tile_cursor,
                                                      uint32_t* __restrict__ tile_queue, uint4* __restrict__ tile_qrec,
                                                      const uint32_t* __restrict__ tile_done, int use_hint,
                                                      uint32_t* __restrict__ chunk_start, MgrHeader* hdr,
                                                      uint32_t cap, uint32_t* __restrict__ tile_zcut,
                                                      uint32_t* __restrict__ tile_zused, uint32_t* __restrict__ tile_qend,
                                                      int use_cut, const uint4* __restrict__ blk_box, DbinArgs db,
                                                      unsigned char* __restrict__ tile_bgok, uint32_t* __restrict__ tile_rep, const MgrRep rep) {
    __shared__ uint32_t s_scan[32];
    if ((int)blockIdx.x >= V * nbT) {   // the workgroups behind the scan's: one per view, depth-bucket offsets + tile box
        __shared__ uint32_t s_box[4];
        __shared__ uint32_t s_items[2 * DBR_MAX_ITEMS];
        dbin_scan_block((int)blockIdx.x - V * nbT, nbT, blk_box, db.db_count, db.db_cursor, db.db_start, db.db_nvis, db.db_bbox,
                        s_scan, s_box, hdr, db.db_item, db.items_per_view, db.items_per_view <= DBR_MAX_ITEMS ? s_items : (uint32_t*)nullptr);
        return;
    }
    __shared__ uint32_t s_gtot[MGR_NCLS], s_gpre[MGR_NCLS], s_vtot[MGR_NCLS], s_vpre[MGR_NCLS];   // tiles per class: all / in front of this block, of all views / of this view
    __shared__ uint32_t s_gbase[MGR_NCLS], s_vbase[MGR_NCLS], s_lc[MGR_NCLS];
    __shared__ uint32_t s_base[2], s_tot[2], s_maxnb, s_nbv;
    const int tid = threadIdx.x, nblk = V * nbT, b = blockIdx.x, v = b / nbT, t = (b % nbT) * 1024 + tid;
    const bool valid = t < T;
    const size_t k = (size_t)v * T + t;
    if (tid < MGR_NCLS) {   // thread c: class c over the blocks (eight loads in flight: the walk is a chain of round trips otherwise)
        uint32_t gt = 0, gp = 0, vt = 0, vp = 0;
        int j0 = 0;
        for (; j0 + 8 <= nblk; j0 += 8) {
            uint32_t x8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x8[u] = blk_cls[(size_t)(j0 + u) * MGR_NCLS + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const uint32_t x = x8[u];
                gt += x;
                if (j < b) gp += x;
                if (j / nbT == v) { vt += x; if (j < b) vp += x; }
            }
        }
        for (int j = j0; j < nblk; ++j) {
            const uint32_t x = blk_cls[(size_t)j * MGR_NCLS + tid];
            gt += x;
            if (j < b) gp += x;
            if (j / nbT == v) { vt += x; if (j < b) vp += x; }
        }
        s_gtot[tid] = gt; s_gpre[tid] = gp; s_vtot[tid] = vt; s_vpre[tid] = vp;
        s_lc[tid] = 0;
    }
    if (tid == 64) {
        uint32_t r = 0, rc = 0, b0 = 0, b1 = 0;
        int j0 = 0;
        for (; j0 + 8 <= nblk; j0 += 8) {
            uint2 p8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p8[u] = part[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u == b) { b0 = r; b1 = rc; }
                r += p8[u].x;
                rc += p8[u].y;
            }
        }
        for (int j = j0; j < nblk; ++j) {
            if (j == b) { b0 = r; b1 = rc; }
            r += part[j].x;
            rc += part[j].y;
        }
        s_base[0] = b0; s_base[1] = b1; s_tot[0] = r; s_tot[1] = rc;
        s_maxnb = 0;
    }
    __syncthreads();
    // non-empty tiles per view (threads 128 ..: one view each, strided) -> the longest view sets the interleaved queue's length
    // (staging phase A's block sums in LDS first -- every thread two or three words, the sums over LDS -- was measured in round 6:
    //  0.018 -> 0.021 ms at eight views, 0.009 -> 0.016 at one; eight loads in flight in the walks above: 0.018 -> 0.013)
    for (int w = tid - 128; w >= 0 && w < V; w += 896) {
        uint32_t nb = 0;
        for (int j = w * nbT; j < (w + 1) * nbT; ++j)
            for (int c = 1; c < MGR_NCLS; ++c) nb += blk_cls[(size_t)j * MGR_NCLS + c];
        atomicMax(&s_maxnb, nb);
    }
    if (tid == 0) {
        uint32_t q = 0, qv = 0;  // descending class order; class 0 (empty tiles) last in the global queue
        for (int c = MGR_NCLS - 1; c >= 0; --c) {
            s_gbase[c] = q;
            q += s_gtot[c];
            s_vbase[c] = qv;
            if (c > 0) qv += s_vtot[c];
        }
        s_nbv = qv;
    }
    __syncthreads();
    const uint32_t c = valid ? tile_count[k] : 0u;
    if (valid) tile_count[k] = 0;  // consumed: left zero for the next forward (no per-call memset of V*T counters)
    uint32_t total, ctotal;
    const uint32_t run = block_excl_scan(c, s_scan, total);
    const uint32_t crun = block_excl_scan(c ? (c - 1) / MGR_CHUNK : 0u, s_scan, ctotal);
    const uint32_t key = mgr_queue_key(c, valid ? tile_done[k] : 0u, use_hint);
    const int cls = key ? 32 - __clz(key) : 0;
    if (valid) {
        tile_start[k] = s_base[0] + run;
        chunk_start[k] = s_base[1] + crun;
        tile_cursor[k] = 0;
        // depth cut: what this forward applied to the tile moves to tile_zused (the blend's check, k_fwd_items), the hint
        // itself is consumed -- k_fwd_items writes the next one for the tiles that have a list
        // (tile_zcut holds either a hint -- the complement of a positive float's bits: top bit set -- or, round 6, a small
        // countdown: the tile ran out of list under a cut a few forwards ago and gets no hint until it has counted down;
        // the countdown stays in place here, k_fwd_items decrements it)
        const uint32_t zraw = tile_zcut[k];
        const bool is_hint = (zraw & 0x80000000u) != 0u;
        const uint32_t zu = (use_cut && is_hint) ? zraw : 0u;
        tile_zused[k] = zu;
        tile_zcut[k] = is_hint ? 0u : zraw;
        tile_rep[k] = 0u;
        // a tile that had a hint saturated in the forward that left it; if nothing at all is listed for it now, the cut may
        // have taken everything the tile should show and no walk will ever notice: flag it here
        if (zu != 0u && c == 0u) { atomicOr(&hdr->acc_flags, MGR_OVF_CUT); atomicOr(&hdr->rep_why, MGR_WHY_EMPTY); }
        tile_qend[k] = 0u;
        if (c != 0u) tile_bgok[k] = 0;      // the blend will write this tile's pixels ("image kept": see BgFill)
    }
    uint32_t rank = 0;
    {   // rank inside the block's tiles of the class: the empty tiles (class 0, most of them) by one atomic per wave + ballot
        const unsigned long long m0 = __ballot(valid && cls == 0);
        if (valid && cls != 0) rank = atomicAdd(&s_lc[cls], 1u);
        if (m0) {
            const int leader = __builtin_ctzll(m0), ln = tid & 63;
            uint32_t base = 0;
            if (ln == leader) base = atomicAdd(&s_lc[0], (uint32_t)__popcll(m0));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (valid && cls == 0) rank = base + (uint32_t)__popcll(m0 & ((1ull << ln) - 1ull));
        }
    }
    if (valid) {
        tile_queue[s_gbase[cls] + s_gpre[cls] + rank] = (uint32_t)k;
        if (cls > 0) {
            const uint32_t st = min(s_base[0] + run, cap), en = min(s_base[0] + run + c, cap);
            tile_qrec[(size_t)(s_vbase[cls] + s_vpre[cls] + rank) * V + v] = make_uint4((uint32_t)k, st, en - st, s_base[1] + crun);
        }
    }
    // holes behind this view's last tile (its blocks share them out)
    for (uint32_t r = s_nbv + (uint32_t)(b % nbT) * 1024u + (uint32_t)tid; r < s_maxnb; r += (uint32_t)nbT * 1024u)
        tile_qrec[(size_t)r * V + v] = make_uint4(MGR_HOLE, 0u, 0u, 0u);
    if (b == 0 && tid == 0) {
        tile_start[(size_t)V * T] = s_tot[0];
        chunk_start[(size_t)V * T] = s_tot[1];
        const uint32_t rect_pairs = hdr->acc_pairs;   // (the per-instance kernel is done; consumed: zero for the next forward)
        hdr->acc_pairs = 0u;
        hdr->total_pairs = rect_pairs;
        hdr->overflow = (s_tot[0] > cap || rect_pairs > cap) ? MGR_OVF_PAIRS : 0u;
        hdr->n_items = 0;
        hdr->item_head = 0;
        hdr->n_rep_units = 0; hdr->rep_list_used = 0; hdr->rep_ck_used = 0;
        hdr->rep = rep;
        hdr->queue_len = s_gbase[0];     // class 0 (empty tiles) starts after all non-empty ones
        hdr->queue_len_i = s_maxnb * (uint32_t)V;
        hdr->queue_small = s_gbase[11];  // classes <= 11: fewer than 2048 pairs
        hdr->queue_giant = s_gbase[13];  // classes >= 14: at least 8192 pairs (split into depth groups before sorting)
        hdr->n_groups = 0;
        hdr->split_head = 0;
        hdr->group_head = 0;
        hdr->queue_head = 0;
        hdr->queue_head3 = s_gbase[11];
        for (int cc = 0; cc < MGR_NCTR; ++cc) hdr->qctr_f[cc * 64] = MGR_FWD_GRID / MGR_NCTR;   // k_blend_fwd: workgroup w starts with queue position w, the tickets behind the grid
    }
}

// ---------------------------------------------------------------------------
// K3: emit one 64-bit key per (Gaussian, tile) pair directly into the tile's
// segment.  Slot within the segment: block base (one returning global atomic per
// (block, non-empty tile)) + LDS rank.  Order inside a segment is arbitrary; the
// per-tile sort on a unique key makes the final order deterministic.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(EMIT_THREADS) void k_emit(int N, int gx, int gy,
                                                      const float* __restrict__ depth,
                                                      const ushort4* __restrict__ rect,
                                                      const unsigned long long* __restrict__ alive,
                                                      const uint32_t* __restrict__ tile_start,
                                                      uint32_t* __restrict__ tile_cursor,
                                                      unsigned long long* __restrict__ keys,
                                                      uint32_t cap, int lds_hist) {
    extern __shared__ uint32_t s_mem[];
    uint32_t* s_hist = s_mem;
    const int v = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * EMIT_THREADS + tid;
    const int T = gx * gy;
    ushort4 rc = make_ushort4(0, 0, 0, 0);
    float z = 0.f;
    unsigned long long am = 0ull;
    if (i < N) {
        rc = rect[(size_t)v * N + i];
        z = depth[(size_t)v * N + i];
        am = alive[(size_t)v * N + i];
    }
    const bool small = (uint32_t)((rc.z - rc.x) * (rc.w - rc.y)) <= 64u;
    const unsigned long long key_hi = ((unsigned long long)__float_as_uint(z)) << 32;
    if (lds_hist) {
        for (int k = tid; k < T; k += EMIT_THREADS) s_hist[k] = 0;
        __syncthreads();
        {
            int k = 0;
            for (int y = rc.y; y < rc.w; ++y)
                for (int x = rc.x; x < rc.z; ++x, ++k)
                    if (!small || ((am >> k) & 1ull)) atomicAdd(&s_hist[y * gx + x], 1u);
        }
        __syncthreads();
        for (int k = tid; k < T; k += EMIT_THREADS) {
            const uint32_t c = s_hist[k];
            if (c) s_hist[k] = tile_start[(size_t)v * T + k] + atomicAdd(&tile_cursor[(size_t)v * T + k], c);
        }
        __syncthreads();
        int k = 0;
        for (int y = rc.y; y < rc.w; ++y)
            for (int x = rc.x; x < rc.z; ++x, ++k) {
                if (small && !((am >> k) & 1ull)) continue;
                const uint32_t pos = atomicAdd(&s_hist[y * gx + x], 1u);
                if (pos < cap) keys[pos] = key_hi | (unsigned)i;
            }
    } else {
        int k = 0;
        for (int y = rc.y; y < rc.w; ++y)
            for (int x = rc.x; x < rc.z; ++x, ++k) {
                if (small && !((am >> k) & 1ull)) continue;
                const size_t vt = (size_t)v * T + y * gx + x;
                const uint32_t pos = tile_start[vt] + atomicAdd(&tile_cursor[vt], 1u);
                if (pos < cap) keys[pos] = key_hi | (unsigned)i;
            }
    }
}

// ---------------------------------------------------------------------------
// K4: per-tile sort.  Persistent workgroups pull tiles (largest first) from the
// queue.  Segments up to SORT_LDS_KEYS keys are sorted in LDS, larger ones in
// place in global memory (L2-resident) by the same network.  The network is the
// "mirror" bitonic form whose compare-exchanges all point the same way, so the
// tail beyond n behaves as +inf without being stored.
// ---------------------------------------------------------------------------
#define SORT_THREADS 1024
#define SORT_LDS_KEYS 8192
#define RS_THREADS 512   // threads of the LDS radix sort (two workgroups share a CU: one's barrier stalls are covered by the other)
#define RS_WAVES (RS_THREADS / 64)

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_mirror(KeyPtr a, uint32_t n, uint32_t npad, int tid,
                                               int nthreads) {
    const uint32_t half = npad >> 1;
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        {   // mirror step: i vs block_end - offset
            const uint32_t hk = k >> 1;
            for (uint32_t t = tid; t < half; t += nthreads) {
                const uint32_t blk = t / hk, off = t - blk * hk;
                const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
                if (l < n) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < half; t += nthreads) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                if (l < n) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

#define SORT_MAX_GROUPS 64
#define SORT_SAMPLES 1024
#define SORT_LDS_EXTRA 2048  // splitters (65 x 8) + counters/offsets (2 x 65 x 4) + cursors

// group index of a key among ascending splitters sp[1..G-1] (sp[g] = first key of group g)
__device__ __forceinline__ int sort_group(const unsigned long long* sp, int G, unsigned long long key) {
    if (G <= 8) {  // few groups (the usual case): count the splitters at or below the key -- uniform LDS addresses, no
        int g = 0;  // data-dependent loop
        for (int j = 1; j < G; ++j) g += key >= sp[j] ? 1 : 0;
        return g;
    }
    int lo = 0, hi = G - 1;  // invariant: key belongs to a group in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (key >= sp[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// LDS radix sort of up to 1024*E (depth, gaussian) keys by 1024 threads, keys held in registers.
// Stable LSD passes over the 32 depth bits, 8 bits per pass; a pass whose digit is the same for
// every key is skipped (depths of one tile share their high bytes).  Key order between passes is
// (wave, row, lane); ranks inside a wave come from wave match masks (8 ballots per row) plus one
// LDS atomic per distinct digit, ranks across waves from a block scan of the 16x256 counters.
// Equal depths keep an arbitrary order through the passes, so a final odd-even sweep orders
// equal-depth neighbours by Gaussian index (the upstream tie-break); it converges immediately
// when there are no exact ties.
// ---------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void lds_radix_sort(unsigned long long (&keys)[E], uint32_t n, unsigned long long* s_keys,
                                               uint32_t* s_cnt /*16*256*/, uint32_t* s_scan /*32*/, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // which passes matter: OR and AND of all depth words
    uint32_t o = 0u, a = ~0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t d = (uint32_t)(keys[r] >> 32);
        if (d != 0xFFFFFFFFu) { o |= d; a &= d; }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
        o |= (uint32_t)__shfl_xor((int)o, sft, 64);
        a &= (uint32_t)__shfl_xor((int)a, sft, 64);
    }
    if (lane == 0) { s_cnt[wave] = o; s_cnt[RS_WAVES + wave] = a; }
    __syncthreads();
    o = 0u; a = ~0u;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) { o |= s_cnt[w]; a &= s_cnt[RS_WAVES + w]; }
    const uint32_t varying = o ^ a;  // bits that differ between some two real keys
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (((varying >> (8 * pass)) & 0xFFu) == 0u) continue;  // uniform across the block
        const int sh = 32 + 8 * pass;
        for (int q = tid; q < RS_WAVES * 256; q += RS_THREADS) s_cnt[q] = 0;
        __syncthreads();
        uint32_t rnk[E / 2];  // two 16-bit in-wave ranks per register
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t d = (uint32_t)(keys[r] >> sh) & 0xFFu;
            unsigned long long m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const int leader = __builtin_ctzll(m);
            uint32_t prev = 0;
            if (lane == leader) prev = atomicAdd(&s_cnt[d * RS_WAVES + wave], (uint32_t)__popcll(m));
            prev = (uint32_t)__shfl((int)prev, leader, 64);
            const uint32_t rk = prev + (uint32_t)__popcll(m & lt);
            if (r & 1) rnk[r >> 1] |= rk << 16;
            else rnk[r >> 1] = rk;
        }
        __syncthreads();
        {   // exclusive scan in (digit major, wave minor) order: 4 counters per thread
            const uint32_t c0 = s_cnt[4 * tid], c1 = s_cnt[4 * tid + 1], c2 = s_cnt[4 * tid + 2], c3 = s_cnt[4 * tid + 3];
            uint32_t tot;
            const uint32_t base = block_excl_scan(c0 + c1 + c2 + c3, s_scan, tot);
            s_cnt[4 * tid] = base;
            s_cnt[4 * tid + 1] = base + c0;
            s_cnt[4 * tid + 2] = base + c0 + c1;
            s_cnt[4 * tid + 3] = base + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t d = (uint32_t)(keys[r] >> sh) & 0xFFu;
            s_keys[s_cnt[d * RS_WAVES + wave] + ((rnk[r >> 1] >> (16 * (r & 1))) & 0xFFFFu)] = keys[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) keys[r] = s_keys[wave * (64 * E) + r * 64 + lane];
        __syncthreads();
    }
    // keys -> LDS in final order (by depth bits; equal depths still in arbitrary order: lds_sort_ties below)
#pragma unroll
    for (int r = 0; r < E; ++r) s_keys[wave * (64 * E) + r * 64 + lane] = keys[r];
    __syncthreads();
}

// Equal-depth neighbours ordered by Gaussian index (the upstream tie-break): odd-even sweeps until nothing moves.
// Exact depth ties are rare, so the caller only comes here after it has seen one while writing the result out.
__device__ __forceinline__ void lds_sort_ties(unsigned long long* s_keys, uint32_t n, int tid) {
    for (;;) {
        int changed = 0;
        for (int phase = 0; phase < 2; ++phase) {
            for (uint32_t i = (uint32_t)phase + 2u * (uint32_t)tid; i + 1 < n; i += 2u * RS_THREADS) {
                const unsigned long long x = s_keys[i], y = s_keys[i + 1];
                if ((x >> 32) == (y >> 32) && x > y) { s_keys[i] = y; s_keys[i + 1] = x; changed = 1; }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(changed)) break;
    }
}

// Sort n <= 16384 keys read from src[0..n) and write the sorted Gaussian ids.
__device__ __forceinline__ void lds_sort_emit(const unsigned long long* __restrict__ src, uint32_t n,
                                              unsigned long long* s_keys, uint32_t* s_cnt, uint32_t* s_scan, int tid,
                                              uint32_t* __restrict__ out) {
    const int lane = tid & 63, wave = tid >> 6;
#define MGR_RADIX_CASE(EE)                                                                       \
    {                                                                                            \
        unsigned long long k[EE];                                                                \
        _Pragma("unroll") for (int r = 0; r < EE; ++r) {                                        \
            const uint32_t p = (uint32_t)(wave * (64 * EE) + r * 64 + lane);                     \
            k[r] = p < n ? src[p] : ~0ull;                                                       \
        }                                                                                        \
        lds_radix_sort<EE>(k, n, s_keys, s_cnt, s_scan, tid);                                    \
    }
    if (n <= 2u * RS_THREADS) MGR_RADIX_CASE(2)
    else if (n <= 4u * RS_THREADS) MGR_RADIX_CASE(4)
    else if (n <= 6u * RS_THREADS) MGR_RADIX_CASE(6)   // (an item of the instance sort is "about 2048 keys": whole buckets, often a few more)
    else if (n <= 8u * RS_THREADS) MGR_RADIX_CASE(8)
    else MGR_RADIX_CASE(16)
#undef MGR_RADIX_CASE
    // write the ids out and look for an out-of-order equal-depth pair on the way (one pass, no extra barrier rounds)
    int tie = 0;
    for (uint32_t t = (uint32_t)tid; t < n; t += (uint32_t)RS_THREADS) {
        const unsigned long long x = s_keys[t];
        out[t] = (uint32_t)x;
        if (t + 1 < n) {
            const unsigned long long y = s_keys[t + 1];
            tie |= ((x >> 32) == (y >> 32) && x > y) ? 1 : 0;
        }
    }
    if (__syncthreads_or(tie)) {
        lds_sort_ties(s_keys, n, tid);
        for (uint32_t t = (uint32_t)tid; t < n; t += (uint32_t)RS_THREADS) out[t] = (uint32_t)s_keys[t];
        __syncthreads();
    }
}

#ifdef MGR_TIMELINE
__device__ unsigned long long g_tl[2048 * 4];
__device__ unsigned long long g_tl2[2048 * 4];
__device__ unsigned long long g_tl3[8192 * 4];
extern "C" int mgr_debug_timeline(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl), sizeof(unsigned long long) * 2048 * 4);
}
extern "C" int mgr_debug_timeline3(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl3), sizeof(unsigned long long) * 8192 * 4);
}
extern "C" int mgr_debug_timeline2(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl2), sizeof(unsigned long long) * 2048 * 4);
}
#endif

// Giant segments (>= SORT_LDS_KEYS pairs): regroup the keys by depth range so that every group fits
// the LDS sort, and hand the groups to k_tile_sort as independent work items — the groups of one
// tile are then sorted by different workgroups instead of one after another.
// Per tile: 1024 sampled keys are sorted, G-1 of them become splitters, one pass counts the group
// sizes and one pass scatters the keys into keys2 (group-contiguous, order inside a group arbitrary).
#define SPLIT_UNROLL 10
__global__ __launch_bounds__(SORT_THREADS) void k_tile_split(
    const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_queue,
    unsigned long long* __restrict__ keys, unsigned long long* __restrict__ keys2,
    uint4* __restrict__ groups, uint32_t* __restrict__ sorted_gid, MgrHeader* hdr, uint32_t cap) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_keys[SORT_SAMPLES];
    __shared__ unsigned long long s_sp[SORT_MAX_GROUPS + 1];
    __shared__ uint32_t s_cnt[SORT_MAX_GROUPS + 1], s_off[SORT_MAX_GROUPS + 1];
    // per-wave group counters, then per-wave write cursors: only lane 0 of the owning wave touches a row, so no
    // atomics (a returning LDS atomic per (trip, group) was the critical path of this kernel) and a fixed key order
    __shared__ uint32_t s_wcnt[SORT_THREADS / 64][SORT_MAX_GROUPS + 1];
    __shared__ uint32_t s_item[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t qlen = hdr->queue_giant;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_item[0] = atomicAdd(&hdr->split_head, 1u);
        __syncthreads();
        const uint32_t item = s_item[0];
#ifdef MGR_TIMELINE
        const unsigned long long tsp0 = wall_clock64();
        if (item >= qlen && tid == 0 && blockIdx.x < 2048) { g_tl2[blockIdx.x * 4 + 2] = tsp0; }
#endif
        if (item >= qlen) break;
        const uint32_t vt = tile_queue[item];
        const uint32_t start = min(tile_start[vt], cap), end = min(tile_start[vt + 1], cap);
        const uint32_t n = end - start;
        if (n == 0) continue;
        const int G = (int)((n + SORT_LDS_KEYS / 2 - 1) / (SORT_LDS_KEYS / 2));
#ifdef MGR_TIMELINE
        unsigned long long tq[6]; tq[0] = wall_clock64();
#define TQ(k) tq[k] = wall_clock64()
#else
#define TQ(k)
#endif
        bool fallback = G > SORT_MAX_GROUPS || n <= SORT_SAMPLES;
        if (!fallback) {
            for (uint32_t t = tid; t < SORT_SAMPLES; t += SORT_THREADS)
                s_keys[t] = keys[start + (uint32_t)(((unsigned long long)t * n) / SORT_SAMPLES)];
            __syncthreads();
            TQ(1);
            bitonic_mirror(s_keys, SORT_SAMPLES, SORT_SAMPLES, tid, SORT_THREADS);
            TQ(2);
            if (tid <= G) s_sp[tid] = (tid == 0) ? 0ull : (tid == G ? ~0ull : s_keys[(tid * SORT_SAMPLES) / G]);
            for (int k = tid; k < (SORT_THREADS / 64) * (SORT_MAX_GROUPS + 1); k += SORT_THREADS) (&s_wcnt[0][0])[k] = 0;
            __syncthreads();
            // both passes over the keys keep SPLIT_UNROLL unconditional loads in flight per thread (clamped
            // index, masked afterwards): one predicated load per trip left the whole 1024-thread workgroup
            // waiting a full memory round trip 2 x n/1024 times
            // The wave's group counters live in a register (lane g holds group g; G <= 64): a read-modify-write of an
            // LDS counter by lane 0 per (key, group) was a chain of dependent LDS round trips -- 26 + 58 us per giant
            // tile in the two passes (per-phase timeline, -DMGR_TIMELINE).
            uint32_t wcnt = 0;
            for (uint32_t t0 = (uint32_t)(tid & ~63); t0 < n; t0 += SORT_THREADS * SPLIT_UNROLL) {  // group sizes
                unsigned long long kk[SPLIT_UNROLL];
#pragma unroll
                for (int u = 0; u < SPLIT_UNROLL; ++u) kk[u] = keys[start + min(t0 + (uint32_t)u * SORT_THREADS + lane, n - 1u)];
#pragma unroll
                for (int u = 0; u < SPLIT_UNROLL; ++u) {
                    const uint32_t t = t0 + (uint32_t)u * SORT_THREADS + lane;
                    const int g = t < n ? sort_group(s_sp, G, kk[u]) : -1;
                    for (int gg = 0; gg < G; ++gg) {
                        const uint32_t c = (uint32_t)__popcll(__ballot(g == gg));
                        wcnt += lane == gg ? c : 0u;
                    }
                }
            }
            if (lane < G) s_wcnt[wave][lane] = wcnt;
            __syncthreads();
            TQ(3);
            if (tid < G) {  // group total; the rows become exclusive prefixes over the waves = each wave's cursor
                uint32_t run = 0;
                for (int w = 0; w < SORT_THREADS / 64; ++w) {
                    const uint32_t c = s_wcnt[w][tid];
                    s_wcnt[w][tid] = run;
                    run += c;
                }
                s_cnt[tid] = run;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t run = 0, big = 0;
                for (int g = 0; g < G; ++g) {
                    s_off[g] = run;
                    run += s_cnt[g];
                    big |= s_cnt[g] > SORT_LDS_KEYS;
                }
                s_item[2] = big;
                s_item[3] = big ? 0u : atomicAdd(&hdr->n_groups, (uint32_t)G);
            }
            __syncthreads();
            fallback = s_item[2] != 0;
        }
        if (!fallback) {
            // lane g: this wave's write cursor inside group g (absolute position in keys2)
            uint32_t wcur = lane < G ? start + s_off[lane] + s_wcnt[wave][lane] : 0u;
            for (uint32_t t0 = (uint32_t)(tid & ~63); t0 < n; t0 += SORT_THREADS * SPLIT_UNROLL) {  // scatter
                unsigned long long kk[SPLIT_UNROLL];
#pragma unroll
                for (int u = 0; u < SPLIT_UNROLL; ++u) kk[u] = keys[start + min(t0 + (uint32_t)u * SORT_THREADS + lane, n - 1u)];
#pragma unroll
                for (int u = 0; u < SPLIT_UNROLL; ++u) {
                    const uint32_t t = t0 + (uint32_t)u * SORT_THREADS + lane;
                    const unsigned long long key = kk[u];
                    const int g = t < n ? sort_group(s_sp, G, key) : -1;
                    for (int gg = 0; gg < G; ++gg) {
                        const unsigned long long mk = __ballot(g == gg);
                        if (!mk) continue;
                        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)wcur, gg);
                        if (g == gg) keys2[base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = key;
                        wcur += lane == gg ? (uint32_t)__popcll(mk) : 0u;
                    }
                }
            }
            if (tid < G) groups[s_item[3] + tid] = make_uint4(start + s_off[tid], s_cnt[tid], 0u, 0u);
#ifdef MGR_TIMELINE
            TQ(4);
            if (tid == 0 && blockIdx.x < 2048) { g_tl2[blockIdx.x * 4 + 0] = tsp0; g_tl2[blockIdx.x * 4 + 1] = wall_clock64(); g_tl2[blockIdx.x * 4 + 3] = n; }
            if (tid == 0 && item < 2048) { g_tl[item * 4 + 0] = tq[1] - tq[0]; g_tl[item * 4 + 1] = tq[2] - tq[1]; g_tl[item * 4 + 2] = tq[3] - tq[2]; g_tl[item * 4 + 3] = ((tq[4] - tq[3]) << 32) | n; }
#endif
        } else {
            // last resort (pathological depth distributions): in-place network in global memory
            uint32_t npad = 1;
            while (npad < n) npad <<= 1;
            __syncthreads();
            bitonic_mirror(keys + start, n, npad, tid, SORT_THREADS);
            __threadfence_block();
            for (uint32_t t = tid; t < n; t += SORT_THREADS) sorted_gid[start + t] = (uint32_t)keys[start + t];
        }
    }
}

// Segments of [2048, 16384) pairs and the depth groups of the giant tiles: one LDS sort each.
__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tile_sort(
    const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_queue,
    const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
    const uint4* __restrict__ groups, uint32_t* __restrict__ sorted_gid, MgrHeader* hdr, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_keys = (unsigned long long*)s_raw;
    uint32_t* s_cnt = (uint32_t*)(s_raw + (size_t)SORT_LDS_KEYS * 8);  // 16 x 256 counters (all LDS in the one array)
    uint32_t* s_scan = s_cnt + RS_WAVES * 256;                          // 32 words
    uint32_t* s_item = s_scan + 32;
    const int tid = threadIdx.x;
    const uint32_t n_groups = hdr->n_groups, q0 = hdr->queue_giant, q1 = hdr->queue_small;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_item[0] = atomicAdd(&hdr->group_head, 1u);
        __syncthreads();
        const uint32_t item = s_item[0];
        const unsigned long long* src;
        uint32_t start, n;
        if (item < n_groups) {  // groups first: they belong to the deepest tiles
            const uint4 gr = groups[item];
            src = keys2; start = gr.x; n = min(gr.y, (uint32_t)SORT_LDS_KEYS);
        } else {
            const uint32_t qi = q0 + (item - n_groups);
            if (qi >= q1) break;
            const uint32_t vt = tile_queue[qi];
            src = keys;
            start = min(tile_start[vt], cap);
            n = min(min(tile_start[vt + 1], cap) - start, (uint32_t)SORT_LDS_KEYS);
        }
        if (n == 0) continue;
        lds_sort_emit(src + start, n, s_keys, s_cnt, s_scan, tid, sorted_gid + start);
    }
}

// Segments of fewer than 2048 pairs (most tiles): 256-thread workgroups with 16 KB of LDS, so
// that many of them are resident per CU.
#define SORT_SMALL_KEYS 2048
__global__ __launch_bounds__(256) void k_tile_sort_small(
    const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_queue,
    const unsigned long long* __restrict__ keys, uint32_t* __restrict__ sorted_gid, MgrHeader* hdr,
    uint32_t cap) {
    __shared__ __attribute__((aligned(16))) unsigned long long s_keys[SORT_SMALL_KEYS];
    __shared__ uint32_t s_item;
    const int tid = threadIdx.x;
    const uint32_t qlen = hdr->queue_len;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_item = atomicAdd(&hdr->queue_head3, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= qlen) break;
        const uint32_t vt = tile_queue[item];
        const uint32_t start = min(tile_start[vt], cap), end = min(tile_start[vt + 1], cap);
        const uint32_t n = min(end - start, (uint32_t)SORT_SMALL_KEYS);
        if (n == 0) continue;
        uint32_t npad = 1;
        while (npad < n) npad <<= 1;
        for (uint32_t t = tid; t < n; t += 256) s_keys[t] = keys[start + t];
        __syncthreads();
        if (n > 1) bitonic_mirror(s_keys, n, npad, tid, 256);
        for (uint32_t t = tid; t < n; t += 256) sorted_gid[start + t] = (uint32_t)s_keys[t];
    }
}

// ---------------------------------------------------------------------------
// K3'/K4': depth-ordered binning (the default route; k_emit + the per-tile sorts above remain as the
// route for images whose tile grid does not fit an LDS histogram, and for A/B runs: MGR_BINNING=sorted).
//
// Every tile list must come out in (depth, Gaussian index) order.  The per-tile sorts order ~R = 5.4 N
// pairs per view; here the N instances of a view are sorted ONCE by (depth, index) and the pairs are then
// *generated in that order*, so that no list needs sorting:
//   1. depth buckets: instances -> 8192 monotone depth buckets per view (counted by the per-instance forward kernel, scanned
//      and scattered by extra workgroups of the two tile-scan launches: pre_tail, dbin_scan_block, dbin_scatter_block)
//      (float bits of z >> 13: 1024 buckets per octave above the 0.2 cull plane);
//   2. k_dbin_sort: LDS radix sorts (lds_sort_emit) of runs of whole buckets, about DB_CHUNK keys each -> db_order;
//   3. k_bin_count: per block of MGR_BIN_BLOCK depth-consecutive instances, the pairs per tile (LDS
//      histogram) -> one row of the (block, tile) matrix;  k_bin_scan: column-wise exclusive scan on top
//      of tile_start -> the row becomes the block's first slot in every tile list;
//   4. k_bin_scatter: one wave per block takes its instances 64 at a time, expands their pairs into an LDS
//      list in instance order and hands out the slots 64 pairs per step from per-tile cursors in LDS, ranking
//      the lanes that share a tile (see the kernel).  LDS operations of one wave complete in program order,
//      which is what makes the steps of a block sequential.  The general fallback walks instance by instance: four instances
//      of at most 16 tiles share a step (16 lanes each, their LDS adds issued one instance after the other).
// The result is bit-identical to the sorted route (unique keys: there is one correct order).
// ---------------------------------------------------------------------------
// Sort the bucketed keys.  Item (view, c) takes the buckets whose first key lies in [c, c + 1) * DB_CHUNK of the view's
// segment: whole buckets, about DB_CHUNK keys, every bucket in exactly one item.  One LDS sort when that is at most
// SORT_LDS_KEYS keys; bucket by bucket otherwise; a single bucket beyond SORT_LDS_KEYS (all Gaussians in one depth
// plane) falls back to the in-place bitonic network in global memory.
#ifndef DBS_WAVES_EU
#define DBS_WAVES_EU 4
#endif
#ifndef DB_CHUNK
#define DB_CHUNK 2048
#endif
// DB_CHUNK:        // (1024 when there are few instances in all: more, shorter items)
__device__ __forceinline__ uint32_t db_lower_bound(const uint32_t* __restrict__ st, uint32_t key) {
    uint32_t lo = 0, hi = MGR_DB_BUCKETS;   // first bucket b in [0, MGR_DB_BUCKETS] with st[b] >= key (st is non-decreasing)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (st[mid] >= key) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// Two launches can share the items by size: the light one (LDS for DBS_LIGHT_KEYS keys: three workgroups per CU instead of
// two, so the ~530 items of the bench step run in one round instead of two) takes the items of at most that many keys and
// larger ones bucket by bucket, all but those with a single bucket beyond its LDS -- those (a dense depth slice: typical
// with one view and rows of 1024 keys) are the full launch's (SORT_LDS_KEYS), or, when the caller skips the full launch
// (debug bit 128), go through the light one's global-memory fallback.  The header counts them (sort_big) so that the caller
// can choose for the next forward: none -> the light launch alone (with the depth cut the instances that take part halve:
// 0.051 -> 0.044 ms at eight views), some -> the full launch alone (one view: 0.039 ms; the light one alone took 0.093 there);
// either way every item is sorted by exactly one launch.  lds_keys: capacity of this launch; min_keys: items of at most this many keys belong to the other launch;
// full_runs: the full launch follows (light launch only).
#define DBS_LIGHT_KEYS 4096
#ifdef DBS_PROF   // tools/instr/dbsprof.py: per-item times of k_dbin_sort (wall_clock64, thread 0)
__device__ unsigned long long g_dbsprof[16];
extern "C" int mgr_debug_dbsprof(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbsprof), sizeof(unsigned long long) * 16);
}
#endif
// Background of the EMPTY tiles (85 % of the tiles of a capture-like frame: 170 MB of stores at eight 1080p views).  The forward
// blend used to write it behind its own work -- 0.04 of its 0.30 ms, measured by leaving the stores out -- while the instance
// sort, which runs when the tile counts are final, keeps 240 workgroups busy for 39 us on a GPU whose memory system idles:
// extra workgroups of that launch write it there.
struct BgFill {
    float* out;                    // (V, 3, H, W); nullptr: no fill blocks in this launch
    const uint32_t* tile_queue;    // non-empty tiles first (hdr->queue_len of them), the empty ones behind
    const float* bg;
    int VT, T, gx, W, H;
    // "image kept" (debug bit 1024): the caller vouches that `out` is the image of the previous complete forward on this
    // workspace, untouched -- a tile that was background then (tile_bgok, kept by k_tile_scan_b / k_fwd_items) and is empty now
    // is left alone.  Honoured when the header says that forward was the last one binned here, wrote THIS buffer, with THIS colour.
    const unsigned char* tile_bgok;
    int kept;
};
__device__ __forceinline__ void bg_fill_block(const BgFill& f, const MgrHeader* hdr, int fb, int nfb) {
    const int tid = threadIdx.x;
    const size_t P = (size_t)f.W * f.H;
    const float b0 = f.bg[0], b1 = f.bg[1], b2 = f.bg[2];
    const uint32_t q0 = hdr->queue_len;
    const unsigned long long owner = (unsigned long long)(uintptr_t)f.out;
    const bool kept_ok = f.kept && hdr->img_seq != 0u && hdr->img_seq + 1u == hdr->fwd_seq && hdr->img_owner[0] == (uint32_t)owner &&
                         hdr->img_owner[1] == (uint32_t)(owner >> 32) && hdr->img_bg[0] == __float_as_uint(b0) &&
                         hdr->img_bg[1] == __float_as_uint(b1) && hdr->img_bg[2] == __float_as_uint(b2);
    if ((f.W & 3) == 0 && (((uintptr_t)f.out) & 15) == 0) {
        // 16-byte stores: a tile is 3 planes x 16 rows x 4 quads = 192 stores; a workgroup of 512 threads takes 8 tiles per step
        // (1536 stores, three per thread); lanes run over (quad, tile) first: the eight tiles of a batch are neighbours in the
        // queue and mostly in the image, so 32 lanes write 512 contiguous bytes of one image row
        auto fill8 = [&](uint32_t qb, uint32_t need8) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = tid + k * RS_THREADS;            // 0 .. 1535
                const int quad = e & 3, ti = (e >> 2) & 7, row = (e >> 5) & 15, plane = e >> 9;
                const uint32_t q = qb + (uint32_t)ti;
                if (q >= (uint32_t)f.VT || !((need8 >> ti) & 1u)) continue;
                const uint32_t vt = f.tile_queue[q];
                const int v = (int)(vt / (uint32_t)f.T), t = (int)(vt % (uint32_t)f.T);
                const int px = (t % f.gx) * 16 + quad * 4, py = (t / f.gx) * 16 + row;
                if (px < f.W && py < f.H) {
                    const float c = plane == 0 ? b0 : plane == 1 ? b1 : b2;
                    *(float4*)(f.out + ((size_t)v * 3 + plane) * P + (size_t)py * f.W + px) = make_float4(c, c, c, c);
                }
            }
        };
        if (!kept_ok) {
            for (uint32_t qb = q0 + 8u * (uint32_t)fb; qb < (uint32_t)f.VT; qb += 8u * (uint32_t)nfb) fill8(qb, 0xFFu);
            return;
        }
        // image kept: 64 tiles are looked at together, one per lane (every wave of the workgroup on its own: the same answer),
        // and only the batches that hold a tile which was not background before are written
        const int lane = tid & 63;
        for (uint32_t qs = q0 + 64u * (uint32_t)fb; qs < (uint32_t)f.VT; qs += 64u * (uint32_t)nfb) {
            const uint32_t q = qs + (uint32_t)lane;
            const bool need = q < (uint32_t)f.VT && f.tile_bgok[f.tile_queue[min(q, (uint32_t)f.VT - 1u)]] == 0;
            const unsigned long long m = __ballot(need);
            if (m == 0ull) continue;
            for (int b8 = 0; b8 < 8; ++b8) {
                const uint32_t need8 = (uint32_t)(m >> (8 * b8)) & 0xFFu;
                if (need8) fill8(qs + 8u * (uint32_t)b8, need8);
            }
        }
        return;
    }
    const int sub = tid >> 8, t8 = tid & 255;      // RS_THREADS = 512: two tiles per step
    for (uint32_t q = q0 + (uint32_t)(2 * fb + sub); q < (uint32_t)f.VT; q += (uint32_t)(2 * nfb)) {
        const uint32_t vt = f.tile_queue[q];
        if (kept_ok && f.tile_bgok[vt]) continue;
        const int v = (int)(vt / (uint32_t)f.T), t = (int)(vt % (uint32_t)f.T);
        const int px = (t % f.gx) * 16 + (t8 & 15), py = (t / f.gx) * 16 + (t8 >> 4);
        if (px < f.W && py < f.H) {
            float* o = f.out + (size_t)v * 3 * P + (size_t)py * f.W + px;
            o[0] = b0;
            o[P] = b1;
            o[2 * P] = b2;
        }
    }
}
template <int LDS_KEYS>   // (a template so that the light instantiation sheds the 16-keys-per-thread case and its spills)
__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(DBS_WAVES_EU, DBS_WAVES_EU))) void k_dbin_sort(
    int N, int chunk, int chunks_per_view, int n_items, const uint32_t* __restrict__ db_start, const uint32_t* __restrict__ db_nvis,
    unsigned long long* __restrict__ db_keys, uint32_t* __restrict__ db_order, uint32_t min_keys, MgrHeader* hdr,
    int full_runs, BgFill fill, int n_sort_blocks) {
    if ((int)blockIdx.x >= n_sort_blocks) {   // (the launch's extra workgroups: see BgFill)
        bg_fill_block(fill, hdr, (int)blockIdx.x - n_sort_blocks, (int)gridDim.x - n_sort_blocks);
        return;
    }
    // full_runs == 2 (round 6): the launch behind k_dbin_rank -- it takes the items of more than min_keys keys, all of them, and
    // returns at once when the ranking kernel met none (MgrHeader::sort_huge: the usual case; sort_big also counts the items from
    // 13/16 of the limit on, which decide whether the caller may skip this launch next time)
    if (full_runs == 2 && hdr->sort_huge == 0u) return;
    constexpr uint32_t lds_keys = (uint32_t)LDS_KEYS;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_keys = (unsigned long long*)s_raw;
    uint32_t* s_cnt = (uint32_t*)(s_raw + (size_t)lds_keys * 8);
    uint32_t* s_scan = s_cnt + RS_WAVES * 256;
    const int tid = threadIdx.x;
#ifdef DBS_PROF
    const long long t_start = wall_clock64();
#endif
    for (int item = blockIdx.x; item < n_items; item += n_sort_blocks) {
#ifdef DBS_PROF
        const long long t0 = wall_clock64();
#endif
        const int v = item % (n_items / chunks_per_view), ch = item / (n_items / chunks_per_view);   // view-minor: neighbours differ in view
        const uint32_t nvis = db_nvis[v];
        if ((uint32_t)ch * (uint32_t)chunk >= nvis) continue;
        const uint32_t* st = db_start + (size_t)v * (MGR_DB_BUCKETS + 1);
        const uint32_t b0 = ch == 0 ? 0u : db_lower_bound(st, (uint32_t)ch * (uint32_t)chunk);
        const uint32_t b1 = db_lower_bound(st, (uint32_t)(ch + 1) * (uint32_t)chunk);
        if (b1 <= b0) continue;
        const uint32_t lo = st[b0], hi = st[b1];
        if (hi == lo) continue;
        if (hi - lo <= min_keys) continue;                          // the light launch's
        if (full_runs != 2 && hi - lo > (uint32_t)DBS_LIGHT_KEYS) {
            // an item beyond the light launch's LDS: the light launch still takes it, bucket by bucket, unless one of its
            // buckets alone is beyond it (that one would go through the bitonic network in global memory: ~90 us) -- such
            // items are the full launch's, and they are what the header counts for the caller's next choice
            uint32_t big = 0;
            for (uint32_t p = (uint32_t)tid; p < b1 - b0; p += (uint32_t)RS_THREADS) big |= (st[b0 + p + 1] - st[b0 + p] > (uint32_t)DBS_LIGHT_KEYS) ? 1u : 0u;
            const bool huge = __syncthreads_or((int)big) != 0;
            if (lds_keys < (uint32_t)SORT_LDS_KEYS) {
                if (huge && tid == 0) atomicAdd(&hdr->sort_big, 1u);
                if (huge && full_runs) continue;
            } else {
                if (huge && min_keys == 0u && tid == 0) atomicAdd(&hdr->sort_big, 1u);   // (the full launch alone keeps the count)
                if (!huge && min_keys != 0u) continue;
            }
        }
        unsigned long long* keys = db_keys + (size_t)v * N;
        uint32_t* out = db_order + (size_t)v * N;
        const uint32_t parts = (hi - lo <= lds_keys) ? 1u : b1 - b0;
#ifdef DBS_PROF
        const long long t1 = wall_clock64();
#endif
        for (uint32_t p = 0; p < parts; ++p) {
            const uint32_t a = parts == 1u ? lo : st[b0 + p], n = (parts == 1u ? hi : st[b0 + p + 1]) - a;
            if (n == 0) continue;
            __syncthreads();
            if (n <= lds_keys) {
                lds_sort_emit(keys + a, n, s_keys, s_cnt, s_scan, tid, out + a);
            } else {
                uint32_t npad = 1;
                while (npad < n) npad <<= 1;
                bitonic_mirror(keys + a, n, npad, tid, RS_THREADS);   // (global memory: __syncthreads orders it for the block)
                for (uint32_t t = tid; t < n; t += RS_THREADS) out[a + t] = (uint32_t)keys[a + t];
            }
        }
#ifdef DBS_PROF
        if (tid == 0) {
            const long long t2 = wall_clock64();
            atomicAdd(&g_dbsprof[0], (unsigned long long)(t1 - t0));      // prologue (item bounds)
            atomicAdd(&g_dbsprof[1], (unsigned long long)(t2 - t1));      // loads + sorts + output
            atomicAdd(&g_dbsprof[2], 1ull);                                // items
            atomicAdd(&g_dbsprof[3], (unsigned long long)(hi - lo));      // keys
            atomicAdd(&g_dbsprof[4], (unsigned long long)parts);
            atomicMax(&g_dbsprof[5], (unsigned long long)(t2 - t0));      // slowest item
            atomicMax(&g_dbsprof[6], (unsigned long long)(hi - lo));      // largest item
            atomicMax(&g_dbsprof[7], (unsigned long long)(t2 - t_start)); // latest end since the workgroup's start
            const uint32_t nk = hi - lo;
            atomicAdd(&g_dbsprof[nk <= 1024u ? 8 : nk <= 2048u ? 9 : nk <= 3072u ? 10 : nk <= 4096u ? 11 : 12], 1ull);
        }
#endif
    }
}

// Instance sort, round 6.  Up to round 5 an item was ~2048 keys of whole depth buckets in an LDS radix sort: 66 us per step at
// eight views for a million keys -- one item's latency (four passes of counters, scans and barriers), the chip idle around it --
// and 45 us of a 0.53 ms step at one view.  With uniform depth buckets (db_bucket) a bucket holds ~130 keys of a narrow depth
// slice, so items can be small -- MGR_DB_ITEM keys (+ the rest of the last bucket): two keys per thread -- and their keys share
// their high bits: the radix sort skips those passes (lds_radix_sort).  One workgroup per item; its first / last bucket come from
// the table the bucket scan left (dbin_scan_block).  Items beyond MGR_DB_RANK_MAX keys (a dense depth slice: every Gaussian in
// one plane) are counted in MgrHeader::sort_big and left to the launch behind.
// (Measured on the way, eight views: items of 512 keys ordered by counting -- a key's position = the number of smaller keys of
// its item -- 0.15 ms, the 64-bit compares run at a quarter of the rate; items of 256 keys through a bitonic network in LDS
// 0.063 ms, throughput-bound at n log^2 n; one view: 0.029 against the radix items' 0.045.)
#define DBR_THREADS RS_THREADS
// CAP (keys of LDS) is a template argument, not a launch parameter: the sort below is compiled for item sizes up to CAP only --
// with a run-time bound its 8 / 16-keys-per-thread cases stay alive and the kernel takes 0.100 instead of 0.042 ms (registers).
template <int CAP>
__global__ __launch_bounds__(DBR_THREADS) void k_dbin_rank(int N, int V, int items_per_view, const uint32_t* __restrict__ db_start,
                                                           const uint32_t* __restrict__ db_item, const unsigned long long* __restrict__ db_keys,
                                                           uint32_t* __restrict__ db_order, MgrHeader* hdr,
                                                           int no_launch_behind, BgFill fill, int n_sort_blocks) {
    constexpr uint32_t cap_keys = (uint32_t)CAP;
    if ((int)blockIdx.x >= n_sort_blocks) {   // (the launch's extra workgroups: see BgFill)
        bg_fill_block(fill, hdr, (int)blockIdx.x - n_sort_blocks, (int)gridDim.x - n_sort_blocks);
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned long long s_keys[CAP];
    __shared__ uint32_t s_cnt[RS_WAVES * 256];
    __shared__ uint32_t s_scan[32];
    const int tid = threadIdx.x;
    const int v = (int)blockIdx.x % V, it = (int)blockIdx.x / V;      // view-minor: neighbours differ in view
    const uint32_t* const ent = db_item + ((size_t)v * items_per_view + it) * 2;
    const uint32_t e0 = ent[0], e1 = ent[1];                          // (uniform: scalar loads)
    if (e0 == 0u) return;                                             // no bucket starts in this item
    const uint32_t* st = db_start + (size_t)v * (MGR_DB_BUCKETS + 1);
    const uint32_t lo = st[~e0], hi = st[e1], n = hi - lo;
    // counted from 13/16 of a limit on (the launch behind is only skipped clear of it: a model under Adam moves a few keys per step
    // between buckets, and the largest items of the bench scene's still model are 1531 .. 1575 keys) -- for BOTH instantiations,
    // whichever this one is: the caller picks the next forward's by sort_large and skips by the count that belongs to it
    if (n > (uint32_t)MGR_DB_RANK_MAX * 13u / 16u && tid == 0) {
        atomicAdd(&hdr->sort_big, 1u);
        if (n > (uint32_t)MGR_DB_RANK_LARGE * 13u / 16u) atomicAdd(&hdr->sort_near_large, 1u);
        if (n > (uint32_t)MGR_DB_RANK_MAX) atomicAdd(&hdr->sort_large, 1u);
    }
    if (n > cap_keys) {
        if (tid == 0) {
            atomicAdd(&hdr->sort_huge, 1u);
            // the caller skipped the launch behind (debug bit 128: the previous forward met no such item): flagged like a skipped
            // binning tier -- the forward is run again with every launch
            if (no_launch_behind) atomicOr(&hdr->overflow, MGR_OVF_TIER);
        }
        return;
    }
    lds_sort_emit(db_keys + (size_t)v * N + lo, n, s_keys, s_cnt, s_scan, tid, db_order + (size_t)v * N + lo);
}

// The (block, tile) matrix only spans the bounding box of a view's non-empty tiles; a view whose box has at most
// BIN_SMALL_TILES tiles is handled by the SMALL instantiation (8 KB of LDS: many workgroups per CU), the others by the
// one with LDS for the whole grid.  Both are launched; each returns at once for the views of the other.
#define BIN_SMALL_TILES 2048
__device__ __forceinline__ bool bin_mine(ushort4 box, bool small_variant) {
    const uint32_t tb = (uint32_t)box.z * (uint32_t)box.w;
    return tb > 0u && (tb <= (uint32_t)BIN_SMALL_TILES) == small_variant;
}

// pairs per (block of depth-consecutive instances, tile of the box); 256 threads take four instances each (the gathers
// of all four in flight together), eight workgroups per CU
#define BCNT_THREADS 256
#define BCNT_PER (MGR_BIN_BLOCK / BCNT_THREADS)
template <bool SMALL>
__global__ __launch_bounds__(BCNT_THREADS) void k_bin_count(int N, int T, int nblk, int bb,
                                                            const uint32_t* __restrict__ db_nvis,
                                                            const ushort4* __restrict__ db_bbox,
                                                            const uint32_t* __restrict__ db_order,
                                                            const ushort4* __restrict__ rect,
                                                            const unsigned long long* __restrict__ alive,
                                                            uint4* __restrict__ db_rec, uint32_t* __restrict__ bin_mat, int xcd_order) {
    extern __shared__ uint32_t s_mem[];
    uint32_t* s_hist = s_mem;
    // xcd_order (1-D grid): XCD-aware order for a multiple of 8 views -- the rectangle / mask gathers below are random over
    // the view's arrays (4.8 MB at 300 k Gaussians, about one XCD's L2): workgroup w runs on XCD w % 8, so XCD x takes
    // the views x, x + 8, ... and keeps their arrays to itself
    const int tid = threadIdx.x;
    int v = blockIdx.y, b = blockIdx.x;
    if (xcd_order) {
        const int q = blockIdx.x >> 3;
        v = (blockIdx.x & 7) + 8 * (q / nblk);
        b = q % nblk;
    }
    const ushort4 box = db_bbox[v];
    if (!bin_mine(box, SMALL)) return;
    const uint32_t nvis = db_nvis[v];
    if ((uint32_t)b * (uint32_t)bb >= nvis) return;
    const int TB = (int)box.z * (int)box.w, bw = box.z;
    uint32_t gid[BCNT_PER];
#pragma unroll
    for (int r = 0; r < BCNT_PER; ++r) {
        const uint32_t p = (uint32_t)b * (uint32_t)bb + r * BCNT_THREADS + tid;
        gid[r] = (r * BCNT_THREADS < bb && p < nvis) ? db_order[(size_t)v * N + p] : 0u;
    }
    ushort4 rc[BCNT_PER];
    unsigned long long am[BCNT_PER];
#pragma unroll
    for (int r = 0; r < BCNT_PER; ++r) {
        rc[r] = rect[(size_t)v * N + gid[r]];
        am[r] = alive[(size_t)v * N + gid[r]];
    }
    for (int k = tid; k < TB; k += BCNT_THREADS) s_hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BCNT_PER; ++r) {
        const uint32_t p = (uint32_t)b * (uint32_t)bb + r * BCNT_THREADS + tid;
        if (r * BCNT_THREADS < bb && p < nvis) {
            // the gathered rectangle and mask, in depth order, for k_bin_scatter (which then reads them coalesced)
            const uint32_t w = rc[r].z - rc[r].x, h = rc[r].w - rc[r].y;
            db_rec[(size_t)v * N + p] = make_uint4((uint32_t)rc[r].x | ((uint32_t)rc[r].y << 16), w | (h << 16), (uint32_t)am[r],
                                                   (uint32_t)(am[r] >> 32));
            const int org = ((int)rc[r].y - (int)box.y) * bw + (int)rc[r].x - (int)box.x;
            if (w * h <= 64u) {
                const float rw = __frcp_rn((float)max(w, 1u));
                unsigned long long m = am[r];
                while (m) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    const uint32_t ty = (uint32_t)(((float)k + 0.5f) * rw), tx = k - ty * w;   // k / w, exact for k < 64
                    atomicAdd(&s_hist[org + (int)ty * bw + (int)tx], 1u);
                }
            } else {
                for (uint32_t y = 0; y < h; ++y)
                    for (uint32_t x = 0; x < w; ++x) atomicAdd(&s_hist[org + (int)y * bw + (int)x], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* row = bin_mat + ((size_t)v * nblk + b) * T;
    for (int k = tid; k < TB; k += BCNT_THREADS) row[k] = s_hist[k];
}

// column-wise exclusive scan over the blocks of a view, starting at the tile's list start.  A workgroup of 1024 threads
// takes BSCAN_COLS = 32 columns; its BSCAN_SEGS = 32 row segments (32 lanes each: two segments per wave, the loads of a
// wave = 32 columns x 2 rows, coalesced) each sum a 32nd of the rows with eight loads in flight, the partial sums meet in
// LDS, and every segment then rewrites its rows as running offsets.
#define BSCAN_COLS 32
#define BSCAN_SEGS 32   // (16 segments with four loads in flight: 11.5 us at eight views, 21 us at one view's 1172 rows of 256 instances)
__global__ __launch_bounds__(BSCAN_COLS * BSCAN_SEGS) void k_bin_scan(int gx, int T, int nblk, int bb, const uint32_t* __restrict__ db_nvis,
                                                                      const ushort4* __restrict__ db_bbox,
                                                                      const uint32_t* __restrict__ tile_start,
                                                                      uint32_t* __restrict__ bin_mat) {
    __shared__ uint32_t s_part[BSCAN_SEGS][BSCAN_COLS];
    const int v = blockIdx.y, col = threadIdx.x & (BSCAN_COLS - 1), seg = threadIdx.x / BSCAN_COLS;
    const int k = blockIdx.x * BSCAN_COLS + col;
    const ushort4 box = db_bbox[v];
    const int TB = (int)box.z * (int)box.w;
    if (blockIdx.x * BSCAN_COLS >= TB) return;
    const bool on = k < TB;
    const int nb = (int)((db_nvis[v] + (uint32_t)bb - 1u) / (uint32_t)bb);
    const int per = (nb + BSCAN_SEGS - 1) / BSCAN_SEGS, b0 = min(seg * per, nb), b1 = min(b0 + per, nb);
    uint32_t* colp = bin_mat + (size_t)v * nblk * T + (on ? k : 0);
    uint32_t sum = 0;
    if (on) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {   // eight loads in flight: the kernel is its dependent round trips
            uint32_t c[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = colp[(size_t)(b + j) * T];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += c[j];
        }
        for (; b < b1; ++b) sum += colp[(size_t)b * T];
    }
    s_part[seg][col] = sum;
    __syncthreads();
    if (!on) return;
    const int ty = k / (int)box.z, tx = k - ty * (int)box.z;
    uint32_t run = tile_start[(size_t)v * T + (size_t)(box.y + ty) * gx + box.x + tx];
    for (int j = 0; j < seg; ++j) run += s_part[j][col];
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = colp[(size_t)(b + j) * T];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            colp[(size_t)(b + j) * T] = run;
            run += c[j];
        }
    }
    for (; b < b1; ++b) {
        const uint32_t c = colp[(size_t)b * T];
        colp[(size_t)b * T] = run;
        run += c;
    }
}

// one wave per block: instances in order, slots from returning LDS adds on the block's per-tile cursors
struct BinRec {
    uint32_t xy, wh, alo, ahi, gid, tiles;   // x0 | y0 << 16 (signed halves, relative to the box), w | h << 16, alive mask, Gaussian, tiles
};
__device__ __forceinline__ int bin_x0(uint32_t xy) { return (int)(short)(xy & 0xFFFFu); }
__device__ __forceinline__ int bin_y0(uint32_t xy) { return (int)xy >> 16; }
__device__ __forceinline__ BinRec bin_load(int N, int v, uint32_t p, uint32_t nvis, ushort4 box,
                                           const uint32_t* __restrict__ db_order, const uint4* __restrict__ db_rec) {
    BinRec r = {0u, 0u, 0u, 0u, 0u, 0u};
    if (p < nvis) {
        const uint4 q = db_rec[(size_t)v * N + p];      // two independent coalesced loads
        r.gid = db_order[(size_t)v * N + p];
        const uint32_t w = q.y & 0xFFFFu, h = q.y >> 16;
        // relative to the box; negative when the rectangle starts outside it (those tiles are null: the box spans the
        // non-null tiles only), hence signed halves
        r.xy = ((uint32_t)((int)(q.x & 0xFFFFu) - (int)box.x) & 0xFFFFu) | ((uint32_t)((int)(q.x >> 16) - (int)box.y) << 16);
        r.wh = q.y;
        r.tiles = w * h;
        r.alo = q.z; r.ahi = q.w;
    }
    return r;
}
// The lane-th lane of a batch that is spread over lanes (k_bin_scatter, round 6): the instance (or the row piece of a rectangle of
// more than 64 tiles) that virtual lane j of the batch stands for.  s_vend: running count of the lanes the 64 instances of the
// batch take; every lane of the wave must call it (the source records travel by lane exchange).
__device__ __noinline__ BinRec bin_spread_lane(uint32_t j, const BinRec cur, const uint32_t* s_vend) {
    BinRec r = {0u, 0u, 0u, 0u, 0u, 0u};
    const bool on = j < s_vend[63];
    int lo = 0, hi = 63;      // the first source lane whose running count exceeds j
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_vend[mid] > j) hi = mid; else lo = mid + 1;
    }
    BinRec q;
    q.xy = (uint32_t)__shfl((int)cur.xy, lo, 64); q.wh = (uint32_t)__shfl((int)cur.wh, lo, 64);
    q.alo = (uint32_t)__shfl((int)cur.alo, lo, 64); q.ahi = (uint32_t)__shfl((int)cur.ahi, lo, 64);
    q.gid = (uint32_t)__shfl((int)cur.gid, lo, 64); q.tiles = (uint32_t)__shfl((int)cur.tiles, lo, 64);
    if (!on) return r;
    if (q.tiles <= 64u) return q;
    const uint32_t piece = j - (lo ? s_vend[lo - 1] : 0u), qw = q.wh & 0xFFFFu, per_row = (qw + 63u) / 64u;
    const uint32_t row = piece / per_row, px = piece - row * per_row, ww = min(64u, qw - 64u * px);
    r.xy = ((uint32_t)(bin_x0(q.xy) + (int)(64u * px)) & 0xFFFFu) | ((uint32_t)(bin_y0(q.xy) + (int)row) << 16);
    r.wh = ww | (1u << 16);
    const unsigned long long ones = ww >= 64u ? ~0ull : ((1ull << ww) - 1ull);
    r.alo = (uint32_t)ones; r.ahi = (uint32_t)(ones >> 32);
    r.gid = q.gid;
    r.tiles = ww;
    return r;
}

// Slots for one batch of 64 instances, instance by instance (the general route: any rectangle size, any grid).
// Four instances of at most 16 tiles share a step (16 lanes each, their LDS adds issued one instance after the other).
__device__ __forceinline__ void bin_batch_by_instance(const BinRec& r, const BinRec* s_rec, uint32_t* s_cur, int bw, int lane,
                                                      uint32_t* __restrict__ sorted_gid, uint32_t cap) {
    const int g = lane >> 4, kq = lane & 15;
    const unsigned long long quad_ok = __ballot(r.tiles <= 16u);    // (empty lanes past nvis: tiles = 0)
    int i = 0;
#pragma unroll 1
    while (i < 64) {
        if ((i & 3) == 0 && ((quad_ok >> i) & 0xFull) == 0xFull) {
            const BinRec q = s_rec[i + g];
            const uint32_t w = q.wh & 0xFFFFu;
            const uint32_t ty = (uint32_t)(((float)kq + 0.5f) * __frcp_rn((float)max(w, 1u)));   // kq / w, exact for kq < 16
            const uint32_t tx = kq - ty * w;
            const unsigned long long am = ((unsigned long long)q.ahi << 32) | q.alo;
            const bool valid = (uint32_t)kq < q.tiles && ((am >> kq) & 1ull);
            uint32_t* cur = &s_cur[valid ? (bin_y0(q.xy) + (int)ty) * bw + bin_x0(q.xy) + (int)tx : 0];
            uint32_t pa = 0, pb = 0, pc = 0, pd = 0;
            if (valid && g == 0) pa = atomicAdd(cur, 1u);
            __builtin_amdgcn_wave_barrier();
            if (valid && g == 1) pb = atomicAdd(cur, 1u);
            __builtin_amdgcn_wave_barrier();
            if (valid && g == 2) pc = atomicAdd(cur, 1u);
            __builtin_amdgcn_wave_barrier();
            if (valid && g == 3) pd = atomicAdd(cur, 1u);
            __builtin_amdgcn_wave_barrier();
            const uint32_t pos = pa | pb | pc | pd;
            if (valid && pos < cap) sorted_gid[pos] = q.gid;
            i += 4;
        } else {
            const BinRec q = s_rec[i];
            if (q.tiles) {
                const uint32_t w = q.wh & 0xFFFFu;
                const unsigned long long am = ((unsigned long long)q.ahi << 32) | q.alo;
                const bool small = q.tiles <= 64u;
                for (uint32_t k = (uint32_t)lane; k < q.tiles; k += 64) {
                    const uint32_t ty = k / w, tx = k - ty * w;
                    if (small && !((am >> k) & 1ull)) continue;
                    const uint32_t pos = atomicAdd(&s_cur[(bin_y0(q.xy) + (int)ty) * bw + bin_x0(q.xy) + (int)tx], 1u);
                    if (pos < cap) sorted_gid[pos] = q.gid;
                }
            }
            __builtin_amdgcn_wave_barrier();
            i += 1;
        }
    }
}

// The usual route (no rectangle of more than 64 tiles in the batch, at most BIN_PAIR_CAP pairs): the batch's pairs are
// expanded into an LDS list in instance order (every lane walks its own alive mask), then taken 64 at a time with all
// lanes busy.  About a dozen tiles per step are hit by more than one lane (measured on the bench scene), so the lanes
// of a tile must be ranked: through a per-tile lane mask in LDS when the box is SMALL; otherwise by one returning add
// per pair in whatever order the hardware serves the lanes, the cursor read back (a pair alone on its tile sees
// cursor == slot + 1), and the shared tiles put right one at a time (ballot of the tile's lanes -> slot = cursor - group
// size + number of lower lanes, i.e. list = instance order).
#define BIN_PAIR_CAP 1024
#ifdef BIN_PROF
__device__ unsigned long long g_binprof[8 * 4096];
#define BP(k) { const long long now_ = wall_clock64(); acc_[k] += now_ - tp_; tp_ = now_; }
#else
#define BP(k)
#endif
// Two waves per block: wave 0 (producer) reads the instances, scans their pair counts and expands the pair list of
// batch i + 1 while wave 1 (consumer) hands out the slots of batch i -- the cursors and masks are only ever touched by
// the consumer, so the order argument is unchanged; the two batches live in the two halves of a double buffer and the
// waves meet at one workgroup barrier per batch.
// Round 5: TWO consumer waves.  The loop was consumer-bound (ranked steps 34 us per block of 1024 instances against 29 us of
// scan + expansion, -DBIN_PROF): the producer now writes the batch's pairs as two lists -- tiles in even rows of the box from the
// front of the buffer, tiles in odd rows from its back, each in instance order -- and wave 1 takes the even rows, wave 2 the odd
// ones: a tile's list is only ever touched by one wave, so the order argument is unchanged.  The pair counts of both lists
// travel through one packed (16 | 16 bit) scan on the DPP network.
#define BIN_SC_THREADS 192
// c_row_comb[w] = sum over j of 2^(2 w j) below 2^64: times a row's w bits = the even rows of a rectangle of width w
__constant__ unsigned long long c_row_comb[65] = {0x0000000000000000ull, 0x5555555555555555ull, 0x1111111111111111ull, 0x1041041041041041ull, 0x0101010101010101ull, 0x1004010040100401ull, 0x1001001001001001ull, 0x0100040010004001ull, 0x0001000100010001ull, 0x0040001000040001ull, 0x1000010000100001ull, 0x0000100000400001ull, 0x0001000001000001ull, 0x0010000004000001ull, 0x0100000010000001ull, 0x1000000040000001ull, 0x0000000100000001ull, 0x0000000400000001ull, 0x0000001000000001ull, 0x0000004000000001ull, 0x0000010000000001ull, 0x0000040000000001ull, 0x0000100000000001ull, 0x0000400000000001ull, 0x0001000000000001ull, 0x0004000000000001ull, 0x0010000000000001ull, 0x0040000000000001ull, 0x0100000000000001ull, 0x0400000000000001ull, 0x1000000000000001ull, 0x4000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull, 0x0000000000000001ull};
#define BIN_SC_FIXED_BYTES (2 * 64 * sizeof(BinRec) + 2 * BIN_PAIR_CAP * 4 + 3 * 64 * 4 + 16 + 66 * 8)
// (bit 31 of s_info[d]: the batch in buffer d is the producer's last one -- the consumers leave the loop behind it)
// MASKS = tiles the per-tile lane masks cover: BIN_MID_TILES (30 KB of LDS, five workgroups per CU -- the boxes of the
// hand scene hold 960-1470 tiles, tools/instr/tile_bbox.py), BIN_SMALL_TILES (36 KB, four per CU) or 0 (no masks: any box).
#define BIN_MID_TILES 1536
__device__ __forceinline__ bool bin_sc_mine(ushort4 box, int masks) {
    const uint32_t tb = (uint32_t)box.z * (uint32_t)box.w;
    if (tb == 0u) return false;
    const int tier = tb <= (uint32_t)BIN_MID_TILES ? BIN_MID_TILES : (tb <= (uint32_t)BIN_SMALL_TILES ? BIN_SMALL_TILES : 0);
    return tier == masks;
}
// SPREAD (round 6, end): the lane-spreading producer is an instantiation of its own, asked for when the previous forward met
// rectangles of more than 64 tiles (MgrHeader::tiers bit 2 -> debug bit 4096).  Without it a batch that holds such a rectangle
// goes instance by instance, as up to round 5: correct for any input, and the usual producer keeps round 5's loop and
// registers (0.094 -> 0.087 ms at eight views of the capture-like set, where no rectangle is that large).
template <int MASKS, bool SPREAD>
__global__ __launch_bounds__(BIN_SC_THREADS) void k_bin_scatter(int N, int T, int nblk, int bb, const uint32_t* __restrict__ db_nvis,
                                                                const ushort4* __restrict__ db_bbox,
                                                                const uint32_t* __restrict__ db_order,
                                                                const uint4* __restrict__ db_rec,
                                                                const uint32_t* __restrict__ bin_mat,
                                                                uint32_t* __restrict__ sorted_gid, uint32_t cap,
                                                                MgrHeader* hdr, int skipped) {
    constexpr bool SMALL = MASKS > 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    BinRec* s_rec = (BinRec*)s_mem;                                     // [2][64] staged instances (by-instance route)
    unsigned long long* s_mask = (unsigned long long*)(s_mem + 2 * 64 * (sizeof(BinRec) / 4));   // SMALL: lane mask per tile
    uint32_t* s_pairs = (uint32_t*)(s_mask + MASKS);                      // [2][CAP] pairs: tile | lane << 16
    uint32_t* s_gid = s_pairs + 2 * BIN_PAIR_CAP;                       // [2][64] Gaussians of the batch
    uint32_t* s_info = s_gid + 2 * 64;                                  // [2] pairs of the batch, or ~0: by-instance route, or BIN_END
    unsigned long long* s_comb = (unsigned long long*)(s_info + 4);     // [66] copy of c_row_comb (a divergent constant load is a vector-memory round trip)
    uint32_t* s_vend = (uint32_t*)(s_comb + 66);                        // [64] producer: running count of the lanes a batch with a rectangle of more than 64 tiles is spread over
    uint32_t* s_cur = s_vend + 64;                                      // cursors of the box's tiles (absolute list slots)
    const int v = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ushort4 box = db_bbox[v];
    if (!bin_sc_mine(box, MASKS)) {
        // the instantiation that is always launched checks the tiers the caller skipped (bit 0: boxes of more than
        // BIN_SMALL_TILES tiles -- k_bin_count<false>, k_bin_scatter<0>; bit 1: k_bin_scatter<BIN_SMALL_TILES>)
        if (MASKS == BIN_MID_TILES && skipped && b == 0 && tid == 0) {
            const uint32_t tb = (uint32_t)box.z * (uint32_t)box.w;
            if ((tb > (uint32_t)BIN_SMALL_TILES && (skipped & 1)) || (tb > (uint32_t)BIN_MID_TILES && tb <= (uint32_t)BIN_SMALL_TILES && (skipped & 2)))
                atomicOr(&hdr->overflow, MGR_OVF_TIER);
        }
        return;
    }
    const uint32_t nvis = db_nvis[v], p0 = (uint32_t)b * (uint32_t)bb;
    if (p0 >= nvis) return;
    const uint32_t p1 = min(p0 + (uint32_t)bb, nvis);
    const int bw = box.z;
    const int nbatch = (int)((p1 - p0 + 63u) / 64u);
    BinRec nxt = {0u, 0u, 0u, 0u, 0u, 0u};
    if (wave == 0) nxt = bin_load(N, v, p0 + lane, nvis, box, db_order, db_rec);
    {
        const int TB = (int)box.z * (int)box.w;
        const uint32_t* row = bin_mat + ((size_t)v * nblk + b) * T;
        int k = tid;
        for (; k + 3 * BIN_SC_THREADS < TB; k += 4 * BIN_SC_THREADS) {   // four loads in flight
            const uint32_t a0 = row[k], a1 = row[k + BIN_SC_THREADS], a2 = row[k + 2 * BIN_SC_THREADS], a3 = row[k + 3 * BIN_SC_THREADS];
            s_cur[k] = a0; s_cur[k + BIN_SC_THREADS] = a1; s_cur[k + 2 * BIN_SC_THREADS] = a2; s_cur[k + 3 * BIN_SC_THREADS] = a3;
        }
        for (; k < TB; k += BIN_SC_THREADS) s_cur[k] = row[k];
        if (SMALL) for (k = tid; k < TB; k += BIN_SC_THREADS) s_mask[k] = 0ull;
        if (tid < 65) s_comb[tid] = c_row_comb[tid];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef BIN_PROF
    long long acc_[6] = {0, 0, 0, 0, 0, 0};
    const long long t00_ = wall_clock64();
    long long tp_ = t00_;
#endif
    // Round 6: rectangles of more than 64 tiles (a third of the Gaussians when the cameras are close to the hand) no longer send
    // their whole batch through the instance-by-instance route (measured on the close-up camera set: k_bin_scatter 1.2 ms of a
    // 4.1 ms step).  A batch that holds one is spread over LANES: every row of such a rectangle -- pieces of at most 64 tiles --
    // becomes a lane of its own with an all-ones mask (nothing of these rectangles is culled), the others keep their lane, in
    // instance order; the lanes go through the expansion below 64 at a time, so the producer emits a data-dependent number of
    // batches and ends the loop with BIN_END.  Pieces of one rectangle cover different tiles: their order among themselves
    // does not matter, and a tile still sees its entries in instance order.
    int src_it = 0;                 // producer: source batches taken
    uint32_t v_total = 0, v_off = 0;   // lanes the current source batch is spread over / already emitted
    bool v_mapped = false;
    BinRec cur = {0u, 0u, 0u, 0u, 0u, 0u};
    bool last_emitted = false;      // producer: the batch emitted in the previous iteration was the last one
    bool met_big = false;           // producer: a rectangle of more than 64 tiles among this block's instances (-> MgrHeader::tiers bit 2)
#pragma unroll 1
    for (int it = 0; SPREAD || it <= nbatch; ++it) {
        __syncthreads();   // batch it - 1 is expanded (and, the first time, the cursors are loaded); batch it - 2 is consumed
        BP(5)
        if (wave == 0) {
            const int d = it & 1;
            if (SPREAD && last_emitted) break;      // (the consumers take the last batch in this iteration and leave by its bit 31: same barrier count)
            if (SPREAD || it < nbatch) {   // ---- producer: the next 64 lanes into buffer it & 1
                BinRec r;
                if constexpr (!SPREAD) {
                    r = nxt;
                    BP(0)
                    if (it + 1 < nbatch) nxt = bin_load(N, v, p0 + 64u * (uint32_t)(it + 1) + lane, nvis, box, db_order, db_rec);
                } else {
                if (v_off >= v_total) {      // the next source batch
                    cur = nxt;
                    BP(0)
                    ++src_it;
                    if (src_it < nbatch) nxt = bin_load(N, v, p0 + 64u * (uint32_t)src_it + lane, nvis, box, db_order, db_rec);
                    const uint32_t cw = cur.wh & 0xFFFFu, ch = cur.wh >> 16;
                    v_mapped = __ballot(cur.tiles > 64u) != 0ull;
                    v_off = 0u;
                    v_total = 64u;
                    if (v_mapped) {
                        met_big = true;
                        const uint32_t nl = cur.tiles == 0u ? 0u : (cur.tiles <= 64u ? 1u : ch * ((cw + 63u) / 64u));
                        const uint32_t ve = mgr_wave_incl_scan_u32(nl);
                        v_total = (uint32_t)__builtin_amdgcn_readlane((int)ve, 63);
                        s_vend[lane] = ve;
                        __builtin_amdgcn_wave_barrier();      // (same wave: LDS operations complete in program order)
                        if (v_total == 0u) v_total = 1u;         // (an empty batch still passes through once)
                    }
                }
                r = cur;
                if (v_mapped) r = bin_spread_lane(v_off + (uint32_t)lane, cur, s_vend);      // (out of line: the usual path keeps its schedule)
                v_off += 64u;
                last_emitted = v_off >= v_total && src_it >= nbatch;      // (travels in bit 31 of s_info[d])
                }
                const unsigned long long am = ((unsigned long long)r.ahi << 32) | r.alo;
                const bool big = !SPREAD && r.tiles > 64u;      // (SPREAD: no lane holds more than 64 tiles any more)
                const bool use = r.tiles != 0u && !big;
                const uint32_t w = r.wh & 0xFFFFu;
                // the alive tiles in even rows of the box: rows of the rectangle are runs of w bits
                unsigned long long amE = 0ull;
                if (use) {   // (no loop over the rows: the lanes' rectangles differ and the wave would run to the tallest)
                    const unsigned long long rb = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
                    const unsigned long long evens = rb * s_comb[min(w, 64u)];   // rows 0, 2, ... of the rectangle
                    // a rectangle that starts on an odd box row has its rows 1, 3, ... on even box rows: the comb moves up one
                    // row.  A row of 64 tiles (w == 64, one row: tiles <= 64) has no second row -- and a shift by 64 is none
                    // at all: on an odd box row such a rectangle belongs to the odd list entirely
                    const unsigned long long rowm = (bin_y0(r.xy) & 1) ? (w >= 64u ? 0ull : evens << w) : evens;
                    amE = am & rowm;
                }
                const unsigned long long amU = use ? am : 0ull;
                const uint32_t cntE = (uint32_t)__popcll(amE), cntO = (uint32_t)__popcll(amU & ~amE);
                const uint32_t incl = mgr_wave_incl_scan_u32(cntE | (cntO << 16));   // (both sums stay below 2^16: 64 lanes x 64 tiles)
                const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t PE = P & 0xFFFFu, PO = P >> 16;
                BP(1)
                const bool any_big = !SPREAD && __ballot(big) != 0ull;
                met_big = met_big || any_big;
                if (!any_big && PE + PO <= (uint32_t)BIN_PAIR_CAP) {
                    // expand: the lane's pairs in row-major order of the rectangle, even box rows at [oE, oE + cntE) from the front,
                    // odd ones at CAP - 1 - [oO, oO + cntO) from the back
                    uint32_t* pl = s_pairs + d * BIN_PAIR_CAP;
                    uint32_t oE = (incl & 0xFFFFu) - cntE, oO = (incl >> 16) - cntO;
                    const float rw = __frcp_rn((float)max(w, 1u));
                    const int org = bin_y0(r.xy) * bw + bin_x0(r.xy);
                    unsigned long long m = amU;
                    while (m) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1ull;
                        const uint32_t ty = (uint32_t)(((float)k + 0.5f) * rw), tx = k - ty * w;   // k / w, exact for k < 64
                        const uint32_t isE = (uint32_t)((amE >> k) & 1ull);
                        const uint32_t at = isE ? oE : (uint32_t)BIN_PAIR_CAP - 1u - oO;
                        oE += isE;
                        oO += 1u - isE;
                        pl[at] = (uint32_t)(org + (int)ty * bw + (int)tx) | ((uint32_t)lane << 16);
                    }
                    s_gid[d * 64 + lane] = r.gid;
                    if (lane == 0) s_info[d] = P | (last_emitted ? 0x80000000u : 0u);
                } else {      // (more pairs than the stage holds: the lanes one after the other -- every one of at most 64 tiles now)
                    s_rec[d * 64 + lane] = r;
                    if (lane == 0) s_info[d] = 0x7FFFFFFFu | (last_emitted ? 0x80000000u : 0u);
                }
                BP(2)
            }
        } else if (it >= 1) {   // ---- consumers: batch it - 1 from buffer (it - 1) & 1; wave 1 the even box rows, wave 2 the odd ones
            const int d = (it - 1) & 1;
            const uint32_t Praw = s_info[d], Pboth = Praw & 0x7FFFFFFFu;      // (bit 31: the producer's last batch)
            if (Pboth != 0x7FFFFFFFu) {
                const bool odd = wave == 2;
                const uint32_t P = odd ? Pboth >> 16 : Pboth & 0xFFFFu;
                const uint32_t* const plist = s_pairs + d * BIN_PAIR_CAP;
                auto pl = [&](uint32_t i) -> uint32_t { return plist[odd ? (uint32_t)BIN_PAIR_CAP - 1u - i : i]; };
                const uint32_t* gl = s_gid + d * 64;
                uint32_t pr_n = lane < P ? pl((uint32_t)lane) : 0u;
                if (SMALL) {
                    // Lanes of a step on the same tile find each other through the tile's 64-bit lane mask (LDS OR, read back):
                    // the lowest lane adds the group's size to the cursor, the others take its result + their rank in the mask.
                    // The returning add of step c is still in flight while the masks of step c + 1 are built and read.
                    uint32_t first_p = 0u, rank_p = 0u, gid_p = 0u, src_p = 0u;   // step c - 1: add result pending
                    bool valid_p = false;
#pragma unroll 1
                    for (uint32_t c = 0; c < P + 64; c += 64) {
                        const bool valid = c + lane < P;
                        const uint32_t pr = pr_n;
                        if (c + 64 < P) pr_n = (c + 64 + lane < P) ? pl(c + 64u + (uint32_t)lane) : 0u;   // next step's pairs
                        const uint32_t t = pr & 0xFFFFu;
                        if (valid) atomicOr(&s_mask[t], 1ull << lane);
                        __builtin_amdgcn_wave_barrier();
                        const unsigned long long mm = valid ? *(volatile unsigned long long*)&s_mask[t] : (1ull << lane);
                        const uint32_t rank = (uint32_t)__popcll(mm & lt);
                        const bool leader = valid && rank == 0u;
                        uint32_t first = 0u;
                        if (leader) {
                            first = atomicAdd(&s_cur[t], (uint32_t)__popcll(mm));
                            s_mask[t] = 0ull;
                        }
                        const uint32_t gid = gl[pr >> 16];
                        // finish step c - 1
                        if (c > 0) {
                            const uint32_t f = (uint32_t)__shfl((int)first_p, (int)src_p, 64);
                            const uint32_t pos = f + rank_p;
                            if (valid_p && pos < cap) sorted_gid[pos] = gid_p;
                        }
                        first_p = first; rank_p = rank; gid_p = gid; valid_p = valid;
                        src_p = (uint32_t)__builtin_ctzll(mm);
                    }
                } else {
#pragma unroll 1
                    for (uint32_t c = 0; c < P; c += 64) {
                        const bool valid = c + lane < P;
                        const uint32_t pr = pr_n;
                        if (c + 64 < P) pr_n = (c + 64 + lane < P) ? pl(c + 64u + (uint32_t)lane) : 0u;   // next step's pairs
                        const uint32_t t = pr & 0xFFFFu;
                        uint32_t pos = 0u, cend = 1u;
                        if (valid) {
                            pos = atomicAdd(&s_cur[t], 1u);
                            cend = *(volatile uint32_t*)&s_cur[t];   // after the adds of every lane of this step (LDS: in order)
                        }
                        const uint32_t gid = gl[pr >> 16];
                        unsigned long long todo = __ballot(valid && cend - pos > 1u);
                        while (todo) {   // wave-uniform: one tile with several lanes per turn
                            const int first = __builtin_ctzll(todo);
                            const uint32_t tstar = (uint32_t)__builtin_amdgcn_readlane((int)t, first);
                            const bool in = valid && t == tstar;
                            const unsigned long long grp = __ballot(in);
                            if (in) pos = cend - (uint32_t)__popcll(grp) + (uint32_t)__popcll(grp & lt);
                            todo &= ~grp;
                        }
                        if (valid && pos < cap) sorted_gid[pos] = gid;
                    }
                }
                BP(3)
            } else if (wave == 1) {
                const BinRec r = s_rec[d * 64 + lane];
                bin_batch_by_instance(r, s_rec + d * 64, s_cur, bw, lane, sorted_gid, cap);
                BP(4)
            }
            if (Praw & 0x80000000u) break;      // that was the producer's last batch
        }
    }
    // (the next forward should ask for -- or stay on -- the spreading instantiation.  Read first: with cameras close to the hand
    //  nearly every block meets one, and nine thousand atomics on one address cost 0.4 ms)
    if (wave == 0 && lane == 0 && met_big && (*(volatile uint32_t*)&hdr->tiers & 4u) == 0u) atomicOr(&hdr->tiers, 4u);
#ifdef BIN_PROF
    {
        const int wid = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && wid < 4096) {   // producer: 0 wait for records, 1 scan, 2 expansion; consumer: 3 steps, 4 by instance; 5 barrier waits (producer's)
            if (wave == 0) { g_binprof[wid * 8 + 0] = acc_[0]; g_binprof[wid * 8 + 1] = acc_[1]; g_binprof[wid * 8 + 2] = acc_[2];
                             g_binprof[wid * 8 + 5] = (unsigned long long)(tp_ - t00_); g_binprof[wid * 8 + 6] = (unsigned long long)t00_;
                             g_binprof[wid * 8 + 7] = (unsigned long long)tp_; }
            else { g_binprof[wid * 8 + 3] = acc_[3]; g_binprof[wid * 8 + 4] = acc_[4]; }
        }
    }
#endif
}

struct FwdRec {
    float4 a, b;
    float c;
};

#ifdef MGR_STATS
__device__ unsigned long long g_fhist[5][65];
extern "C" int mgr_debug_fhist(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fhist), sizeof(unsigned long long) * 5 * 65);
}
#define FH(r, c, v) do { const int c_ = (c); const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_fhist[r][c_], v_); } while (0)
#endif


#ifdef FWD_PROF
// wall_clock64 (100 MHz) per wave and phase of k_blend_fwd: 0 ticket + tile prologue, 1 box test / compaction (+ waiting for
// the batch's records), 2 pair loop, 3 checkpoint + loop overhead, 4 waiting for the tile's other quadrants, 5 tile
// epilogue (image stores, items); 6 tiles, 7 waves
__device__ unsigned long long g_fprof[8];
extern "C" int mgr_debug_fprof(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fprof), sizeof(unsigned long long) * 8);
}
#ifndef FWD_PROF_MIN
#define FWD_PROF_MIN 0u
#endif
#ifndef FWD_PROF_MAX
#define FWD_PROF_MAX 0xFFFFFFFFu
#endif
#define FP(k) { const long long now_ = wall_clock64(); if (fp_on_) facc_[k] += now_ - ftp_; ftp_ = now_; }
#else
#define FP(k)
#endif
// (occupancy of the forward blend: 78 VGPRs = six waves per SIMD as compiled; forced to five or four the kernel and the step
// time stay within the run-to-run noise, 0.369-0.377 ms / 598-607 iters/s on one box)
// (round 5, on 6144 workgroups: 95 VGPRs = five waves per SIMD as compiled 0.2495 ms; forced to four 0.2415, to three 0.2750, to
// six 0.2762 -- four it is; composite and one view unchanged)
#ifndef FWD_WAVES_EU
#define FWD_WAVES_EU 4
#endif
#define FWD_OCC __attribute__((amdgpu_waves_per_eu(FWD_WAVES_EU, FWD_WAVES_EU)))
// ---------------------------------------------------------------------------
// K5, wave-granular: the same walk, but nothing in the kernel waits at a workgroup barrier.  -DFWD_PROF on the workgroup
// version above showed why: per tile of fewer than 1024 list entries (63 % of the tiles of the bench scene) a wave spent
// 15.5 us walking and 33 us around it -- two workgroup barriers that drain the image stores and wait for thread 0's two
// returning atomics (queue ticket, backward-item base: both on one address each, served one after the other device-wide),
// the slowest of the four quadrants, the dependent prologue loads.
//
// A first version handed (tile, quadrant) units to any wave of the device: no waiting at all, and slower (0.393 against
// 0.374 ms) -- the four quadrants of a tile gather the same Gaussian records, and on four CUs of four XCDs each of them pulls
// the lines from memory itself (the counters of the workgroup version: 74 % of the gathers hit the L1 of the CU the four
// waves share, the L2 serves only 27 % of the rest).  So the tile stays on ONE workgroup, but its waves are not tied to a
// quadrant:
//  * a workgroup walks a SEQUENCE of tiles (step 0 = its own index in the queue, later steps = tickets of MgrQueue);
//  * a wave that is free takes the next unclaimed quadrant of that sequence with one LDS atomic (claim c = step 4 c / 4,
//    quadrant c % 4): a wave whose quadrant ends early goes on to the next tile, the slowest quadrant delays nobody;
//  * the wave that claims quadrant 0 of step s draws the ticket of step s + 1, reads the answer after its first batch and
//    publishes the tile's queue record in an LDS slot; whoever claims a quadrant of step s + 1 finds it there;
//  * the epilogue is stores only: image, n_contrib, and the list depth the quadrant consumed (tile_qdone);
//  * the backward's work items, which the workgroup version appended behind a same-address atomic per tile, are built by
//    k_fwd_items after the blend from the four depths of every tile.
// ---------------------------------------------------------------------------
#ifdef MGR_TIMELINE
// -DMGR_TIMELINE: one record per (tile, quadrant) unit of k_blend_fwd: start, end (wall_clock64), list length | quadrant << 28,
// depth consumed; mgr_debug_timeline_w copies them out (tools/instr/timeline_w.py)
#define TLW_PER_WAVE 24
__device__ unsigned long long g_tlw[MGR_FWD_GRID * 4 * TLW_PER_WAVE * 4];   // (a wave's own rows: no atomics in the measured kernel)
extern "C" int mgr_debug_timeline_w(void* dst, unsigned int* n) {
    *n = MGR_FWD_GRID * 4 * TLW_PER_WAVE;
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tlw), sizeof(g_tlw));
    return (int)e;
}
#endif
// per-lane select by a lane mask held in scalar registers (v_cndmask with an SGPR-pair condition)
__device__ __forceinline__ float mgr_sel(unsigned long long m, float a, float b) {
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t mgr_selu(unsigned long long m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
// One pair step of the front-to-back walk (two list entries at one pixel per lane); true when every pixel of the quadrant has
// stopped.  ONE definition for the forward blend and for the repair of depth-cut tiles (k_repair_blend), which must continue a
// walk bit for bit as the forward blend would have.
// Per entry: a = alpha if kept and the pixel still accumulates, else 0; the walk of a pixel ends where T (1 - a) < 1e-4 (that
// entry contributes nothing).  The lane masks of the decisions live in scalar registers: "contributes" = kept & not ended &
// not ending here is formed there, one select per use.
__device__ __forceinline__ bool mgr_fwd_pair_step(const float4& R0, const float4& R1, const float4& R2, const float4& R3, const float4& R4,
                                                  const mgr_v2f fpx2, const mgr_v2f fpy2, float& Tr, mgr_v2f& C01, float& C2, uint32_t& last,
                                                  unsigned long long& done_m, const unsigned long long exec_m) {
    mgr_v2f al;
    unsigned long long ma, mb;
    mgr_pair_alpha_masks(R0, R1, R2, fpx2, fpy2, al, ma, mb);
    {   // entry a
        const unsigned long long keep = ma & ~done_m;
        const float a = mgr_sel(keep, al.x, 0.0f);
        const float testT = Tr * (1.0f - a);
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(testT < 0.0001f);  // a == 0 leaves testT = Tr >= 1e-4
        const unsigned long long contrib = keep & ~stop;
        const float w = mgr_sel(contrib, a * Tr, 0.0f);
        C01 += mgr_v2f{R3.x, R3.y} * w;
        C2 += R4.x * w;
        Tr = mgr_sel(stop, Tr, testT);
        last = mgr_selu(contrib, __float_as_uint(R4.z), last);
        done_m |= stop;
    }
    {   // entry b
        const unsigned long long keep = mb & ~done_m;
        const float a = mgr_sel(keep, al.y, 0.0f);
        const float testT = Tr * (1.0f - a);
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(testT < 0.0001f);
        const unsigned long long contrib = keep & ~stop;
        const float w = mgr_sel(contrib, a * Tr, 0.0f);
        C01 += mgr_v2f{R3.z, R3.w} * w;
        C2 += R4.y * w;
        Tr = mgr_sel(stop, Tr, testT);
        last = mgr_selu(contrib, __float_as_uint(R4.w), last);
        done_m |= stop;
    }
    return (~done_m & exec_m) == 0ull;
}
#define FWD_SLOTS 64   // LDS ring of published steps; a reader spins on its slot from the moment it claims, so it cannot be lapped
__global__ __launch_bounds__(256) FWD_OCC void k_blend_fwd(int N, int W, int H, int gx, int gy, int VT,
                                                     const float* __restrict__ bg,
                                                     const uint32_t* __restrict__ tile_queue,
                                                     const uint4* __restrict__ tile_qrec,
                                                     const uint32_t* __restrict__ sorted_gid,
                                                     const MgrGRec* __restrict__ grec,
                                                     float* __restrict__ out_color,
                                                     uint32_t* __restrict__ n_contrib,
                                                     uint32_t* __restrict__ tile_done,
                                                     uint32_t* __restrict__ tile_qdone,
                                                     float4* __restrict__ ckpt, MgrHeader* hdr,
                                                     const uint32_t* __restrict__ tile_zused, uint32_t* __restrict__ tile_qend, int skip_fill) {
    __shared__ __align__(16) float s_pair[4][32][MGR_PAIR_FLOATS];
    __shared__ __align__(16) uint4 s_qrec[FWD_SLOTS];     // queue record of a published step
    __shared__ uint32_t s_step[FWD_SLOTS];                // which step the slot holds (published last: the flag the readers poll)
    __shared__ uint32_t s_claim;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = gx * gy;
    const uint32_t n_busy = hdr->queue_len;           // non-empty tiles (the empty ones follow them in tile_queue)
    const uint32_t n_queue = hdr->queue_len_i;         // positions of the blend's queue (tile_qrec: interleaved by view, with holes)
    const size_t P = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const unsigned long long lt = (1ull << lane) - 1ull;
    float* const slab = &s_pair[wave][0][0];
    const uint32_t gid_max = (uint32_t)(N > 0 ? N - 1 : 0);

    if (tid < FWD_SLOTS) s_step[tid] = 0xFFFFFFFFu;
    if (tid == 0) s_claim = 0;
    __syncthreads();   // (the only workgroup barrier of the kernel)
#ifdef FWD_PROF
    long long facc_[6] = {0, 0, 0, 0, 0, 0}, ftp_ = wall_clock64(), fnt_ = 0;
    bool fp_on_ = true;
#endif
#ifdef MGR_TIMELINE
    unsigned int tlw_n = 0;
    if (lane == 0) for (int k = 0; k < TLW_PER_WAVE; ++k) g_tlw[(((size_t)blockIdx.x * 4 + wave) * TLW_PER_WAVE + k) * 4 + 1] = 0;
#endif
    MgrQueue queue;
    queue.init(hdr->qctr_f, n_queue, (int)blockIdx.x);
    const uint32_t END = 0xFFFFFFFFu;
    // A wave in a short list claims its NEXT quadrant when it starts (c_next) and, once the step's record is published,
    // fetches that tile's first list indices during the current walk: the next unit then starts one dependent round trip
    // (the Gaussian records) after this one instead of two.  Long lists claim when they end (an early claim would hold a
    // quadrant of the next tile back for as long as this walk takes).
    uint32_t c_next = END;
    bool have_gid = false;
    uint32_t pf_g0 = 0, pf_g1 = 0;
    auto claim = [&]() -> uint32_t {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&s_claim, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    };
    auto slot_read = [&](uint32_t st) -> uint4 {
        uint4 q = s_qrec[st % FWD_SLOTS];
        q.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.x);
        q.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y);
        q.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z);
        q.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w);
        return q;
    };
    while (true) {
        // ---- the next quadrant of the workgroup's tile sequence ----
        const uint32_t c = c_next != END ? c_next : claim();
        c_next = END;
        const uint32_t step = c >> 2, quad = c & 3u;
#ifdef MGR_TIMELINE
        const unsigned long long tlw0 = wall_clock64();
#endif
        uint4 qrec;
        if (step == 0) {   // the first tile of a workgroup is its own index (k_tile_scan_b starts the counters behind the grid)
            qrec = blockIdx.x < n_queue ? tile_qrec[blockIdx.x] : make_uint4(END, 0u, 0u, 0u);
#ifdef FWD_KO_DEEP   // (instrumentation, WRONG RESULTS: the first FWD_KO_DEEP queue positions -- the deepest tiles -- are not walked)
            if (blockIdx.x < FWD_KO_DEEP && qrec.x != END) qrec.x = MGR_HOLE;
#endif
        } else {
            while (__hip_atomic_load(&s_step[step % FWD_SLOTS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != step) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            qrec = slot_read(step);
        }
        if (qrec.x == END) break;
        const bool got_gid = have_gid;
        have_gid = false;
        // quadrant 0 of a step provides the next step: ticket now, answer + queue record after the first batch
        const bool provider = quad == 0u;
        uint32_t raw = 0;
        if (provider) raw = queue.issue(lane);
        bool provided = !provider;

        const bool hole = qrec.x == MGR_HOLE;   // (interleaved queue: a position its view has no tile for; still provides the next step)
        const uint32_t vt = hole ? 0u : qrec.x;
        const int v = (int)(vt / (uint32_t)T), t = (int)(vt % (uint32_t)T);
        const int bx = t % gx, by = t / gx;
        const uint32_t start = qrec.y, nlist = qrec.z;
        const int px = bx * 16 + (int)(quad & 1u) * 8 + (lane & 7);
        const int py = by * 16 + (int)(quad >> 1) * 8 + (lane >> 3);
        const bool inside = px < W && py < H && !hole;
        const mgr_v2f fpx2 = {(float)px, (float)px}, fpy2 = {(float)py, (float)py};
        const float qx0 = (float)(bx * 16 + (int)(quad & 1u) * 8), qy0 = (float)(by * 16 + (int)(quad >> 1) * 8);
        const uint32_t ck0 = qrec.w;  // checkpoint c (c >= 1) of this tile lives at ck0 + c - 1
        const int pslot = (int)(quad << 6) | lane;  // pixel slot inside a checkpoint (same mapping in backward)
        // the kernel ends when the deepest lists end: long lists issue at raised priority, the rest fill the gaps
        if (nlist >= 8192u) __builtin_amdgcn_s_setprio(3);
        else if (nlist >= 2048u) __builtin_amdgcn_s_setprio(2);
        else if (nlist >= 512u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        float Tr = 1.0f, C2 = 0.f;
        mgr_v2f C01 = {0.f, 0.f};   // (red, green) prefix colour
        uint32_t last = 0;
        const unsigned long long exec_m = __builtin_amdgcn_ballot_w64(true);
        unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);   // pixels whose walk has ended (lane mask, scalar)
        const MgrGRec* const gv = grec + (size_t)v * N;

        // software pipeline: rec = record of batch k, gid_n = index of batch k+1.  All pipeline loads are unconditional
        // (indices clamped into the list): a predicated load makes hipcc drain the whole memory queue every batch.
        // Two batches of records in flight (round 4): the per-unit timeline shows the kernel ending with the walks of its
        // deepest lists -- 10 500 consumed entries at 30 ns each from t = 0 to the last microsecond, the chip four fifths
        // empty for the last third -- and a walk alone on its SIMD is one HBM round trip per batch when only the next batch
        // is on its way (its own pair loop is shorter than that once most pixels of the quadrant have stopped).
        FwdRec rec;
        uint32_t gid_n;
        const uint32_t* const sg = sorted_gid + start;
        const uint32_t lastidx = (nlist ? nlist : 1u) - 1u;   // (a list clipped to nothing by the pair capacity: no walk, indices clamped)
#ifndef FWD_PF1
        FwdRec rec1;   // records of batch k + 1 (in flight); gid_n then holds the indices of batch k + 2
#endif
        {
            const uint32_t g0 = min(got_gid ? pf_g0 : sg[min((uint32_t)lane, lastidx)], gid_max);
            gid_n = got_gid ? pf_g1 : sg[min(64u + lane, lastidx)];
            const MgrGRec* r = gv + g0;
            rec.a = *(const float4*)r;
            rec.b = *((const float4*)r + 1);
            rec.c = r->b;
#ifndef FWD_PF1
            const MgrGRec* r1 = gv + min(gid_n, gid_max);
            gid_n = sg[min(128u + lane, lastidx)];
            rec1.a = *(const float4*)r1;
            rec1.b = *((const float4*)r1 + 1);
            rec1.c = r1->b;
#endif
        }
        const bool early = nlist < 2048u;
        if (early) c_next = claim();
        bool pf_tried = !early;
        // the claimed next unit: its queue record if the step is published by now, and its first list indices
        auto prefetch_next = [&]() {
            pf_tried = true;
            const uint32_t ns = c_next >> 2;
            uint4 q;   // (read again from the slot when the unit starts: fewer scalars live across the walk)
            if (ns == 0) {
                if (blockIdx.x >= n_queue) return;
                q = tile_qrec[blockIdx.x];
            } else {
                if (__hip_atomic_load(&s_step[ns % FWD_SLOTS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ns) return;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                q = slot_read(ns);
            }
            if (q.x == END || q.x == MGR_HOLE) return;
            const uint32_t nl1 = (q.z ? q.z : 1u) - 1u;
            pf_g0 = sorted_gid[q.y + min((uint32_t)lane, nl1)];
            pf_g1 = sorted_gid[q.y + min(64u + lane, nl1)];
            have_gid = true;
        };
#ifdef FWD_PROF
        fp_on_ = nlist >= FWD_PROF_MIN && nlist < FWD_PROF_MAX;
        if (fp_on_) ++fnt_;
        if (rec.c == 12345.678f) break;   // (forces the first records to be here before the clock is read)
#endif
        FP(0);
        uint32_t off = 0;
        for (; off < nlist; off += 64) {
            // bounding box of this quadrant's pixels that are still accumulating
            int bx0, by0, bx1, by1;
            if (!mgr_quad_bbox(~done_m & exec_m, bx0, by0, bx1, by1)) break;
            bool alive = false;
            if (off + lane < nlist)
                alive = !mgr_box_dead(rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, mgr_qmax(rec.b.y), qx0 + (float)bx0,
                                      qy0 + (float)by0, qx0 + (float)bx1, qy0 + (float)by1);
            const unsigned long long m = __ballot(alive);
            const int cnt = __popcll(m);
            if (alive) {
                const int rank = __popcll(m & lt);
                float* pb = slab + (rank >> 1) * MGR_PAIR_FLOATS;
                mgr_pair_store<true>(pb, rank & 1, rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, rec.b.y, rec.b.z, rec.b.w, rec.c,
                               off + (uint32_t)lane + 1u);  // 1-based list position
                if ((cnt & 1) && rank == cnt - 1) mgr_pair_pad<true>(pb);
            }
            // issue the gathers of the following batches; they complete during the blend below
#ifdef FWD_PF1
            {
                const MgrGRec* r = gv + min(gid_n, gid_max);
                rec.a = *(const float4*)r;
                rec.b = *((const float4*)r + 1);
                rec.c = r->b;

                gid_n = sg[min(off + 128u + lane, lastidx)];
            }
#else
            FwdRec rec2;
            {
                const MgrGRec* r = gv + min(gid_n, gid_max);
                rec2.a = *(const float4*)r;
                rec2.b = *((const float4*)r + 1);
                rec2.c = r->b;
                gid_n = sg[min(off + 192u + lane, lastidx)];
            }
#endif
            if (provided && !pf_tried) prefetch_next();
            if (!provided) {   // the ticket drawn at the start of the unit is back by now: publish the next step
                const uint32_t nxt = queue.resolve(raw, lane);
                uint4 nrec = make_uint4(END, 0u, 0u, 0u);
                if (nxt != END) nrec = tile_qrec[nxt];
#ifdef FWD_KO_DEEP
                if (nxt != END && nxt < FWD_KO_DEEP) nrec.x = MGR_HOLE;
#endif
                if (lane == 0) {
                    s_qrec[(step + 1u) % FWD_SLOTS] = nrec;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&s_step[(step + 1u) % FWD_SLOTS], step + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                provided = true;
            }
            const int npair = (cnt + 1) >> 1;
            FP(1);
            // one pair step; true when every pixel of the quadrant has stopped
            auto pair_step = [&](const float4& R0, const float4& R1, const float4& R2, const float4& R3, const float4& R4) -> bool {
                return mgr_fwd_pair_step(R0, R1, R2, R3, R4, fpx2, fpy2, Tr, C01, C2, last, done_m, exec_m);
            };
#ifdef FWD_LDS_PIPE1
            for (int p = 0; p < npair; ++p) {
                const float4* pp = (const float4*)(slab + p * MGR_PAIR_FLOATS);
                const float4 R0 = pp[0], R1 = pp[1], R2 = pp[2], R3 = pp[3], R4 = pp[4];   // (five ds_read_b128 with the (r, g)-per-entry layout)
                if (pair_step(R0, R1, R2, R3, R4)) break;
            }
#else
            // The three records a pair step needs first (positions and conics) are read one step ahead: a walk that is alone
            // on its SIMD -- the deepest lists, which the kernel ends with -- otherwise waits an LDS round trip at the top
            // of every step (12 more VGPRs: 83 -> 95, still five waves per SIMD).  Unrolled by two so that the two sets swap
            // roles without moves.
            if (npair > 0) {
                const float4* pa = (const float4*)slab;
                float4 A0 = pa[0], A1 = pa[1], A2 = pa[2];
                for (int p = 0;; p += 2) {
                    const float4* pb = (const float4*)(slab + min(p + 1, 31) * MGR_PAIR_FLOATS);
                    const float4 B0 = pb[0], B1 = pb[1], B2 = pb[2];
                    const float4* pc = (const float4*)(slab + p * MGR_PAIR_FLOATS);
                    const float4 A3 = pc[3], A4 = pc[4];
                    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the reads to their first use)
                    if (pair_step(A0, A1, A2, A3, A4) || p + 1 >= npair) break;
                    const float4* pn = (const float4*)(slab + min(p + 2, 31) * MGR_PAIR_FLOATS);
                    A0 = pn[0]; A1 = pn[1]; A2 = pn[2];
                    const float4 B3 = pb[3], B4 = pb[4];
                    __builtin_amdgcn_sched_barrier(0);
                    if (pair_step(B0, B1, B2, B3, B4) || p + 2 >= npair) break;
                }
            }
#endif
            FP(2);
            // pixel state in front of the next chunk (prefix colour + transmittance): lets the
            // backward pass process every MGR_CHUNK-entry chunk of the list independently
            const uint32_t nextpos = off + 64u;
            // (only for the pixels that walk on: one that has stopped has its last contributor in front of the next chunk, and the
            // backward reads a chunk's checkpoint for the pixels whose last contributor lies in or behind it -- the checkpoints
            // were 185 MB of stores per bench step and 0.026 ms of this kernel, found by leaving them out)
            if ((nextpos % MGR_CHUNK) == 0 && nextpos < nlist && (((~done_m & exec_m) >> lane) & 1ull))
                ckpt[(size_t)(ck0 + nextpos / MGR_CHUNK - 1) * 256 + pslot] = make_float4(C01.x, C01.y, C2, Tr);
#ifndef FWD_PF1
            rec = rec1;
            rec1 = rec2;
#endif
            FP(3);
        }
        FP(3);
        if (!provided) {   // (the walk ended before its first batch: all pixels outside the image, or an empty list)
            const uint32_t nxt = queue.resolve(raw, lane);
            uint4 nrec = make_uint4(END, 0u, 0u, 0u);
            if (nxt != END) nrec = tile_qrec[nxt];
#ifdef FWD_KO_DEEP
            if (nxt != END && nxt < FWD_KO_DEEP) nrec.x = MGR_HOLE;
#endif
            if (lane == 0) {
                s_qrec[(step + 1u) % FWD_SLOTS] = nrec;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&s_step[(step + 1u) % FWD_SLOTS], step + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (!pf_tried) prefetch_next();
        if (inside) {
            const size_t pix = (size_t)py * W + px;
            n_contrib[(size_t)v * P + pix] = last;  // (the final transmittance is not needed by the chunk-parallel backward)
            float* o = out_color + (size_t)v * 3 * P + pix;
            o[0] = C01.x + Tr * bg0;
            o[P] = C01.y + Tr * bg1;
            o[2 * P] = C2 + Tr * bg2;
        }
        // list depth this quadrant consumed; the tile's maximum drives the backward pass (k_fwd_items)
        const uint32_t mx = mgr_wave_max_u32(last);   // (DPP + readlane: the six __shfl_xor steps were ds_bpermute round trips at the end of every unit)
        // depth cut: the list position in front of which every pixel of the quadrant had stopped (a whole number of
        // batches; the tile's maximum becomes the next forward's hint), or "never" -- and if the list ran out under an
        // unsaturated pixel while this forward had cut it short, the image may lack contributions: the quadrant is handed
        // to the repair kernels with its pixels' state (MgrRepUnit), or, without them, the forward is flagged
        const bool unsat = (~done_m & exec_m) != 0ull;
        if (lane == 0 && !hole) {
            tile_qdone[(size_t)vt * 4 + quad] = mx;
            atomicMax(&tile_qend[vt], unsat ? 0xFFFFFFFFu : min(off, nlist));
        }
        if (unsat && !hole && tile_zused[vt] != 0u) {      // (wave-uniform: rare)
            const MgrRep rep = hdr->rep;                    // (k_tile_scan_b left it there)
            uint32_t u = 0xFFFFFFFFu;
            if (rep.max_units) {
                if (lane == 0) u = atomicAdd(&hdr->n_rep_units, 1u);
                u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
            }
            if (u < rep.max_units) {
                rep.state[(size_t)u * 64 + lane] = make_float4(C01.x, C01.y, C2, Tr);
                rep.last[(size_t)u * 64 + lane] = last;
                if (lane == 0) {
                    MgrRepUnit r;
                    r.vt = vt; r.quad = quad; r.nlist = nlist; r.start = start;
                    r.ck0 = ck0; r.done_lo = (uint32_t)done_m; r.done_hi = (uint32_t)(done_m >> 32); r.ntail = 0u;
                    r.ov_start = 0u; r.ov_ck0 = 0u; r.p0 = 0u; r.ck_first = 0u;
                    r.beyond = 0u; r.pad[0] = r.pad[1] = r.pad[2] = 0u;
                    rep.unit[u] = r;
                    rep.cnt[u] = 0u;
                    atomicMax(&rep.tile_rep[vt], u + 1u);      // the tile's owner unit: the highest of its (up to four) units
                }
            } else if (lane == 0) {
                atomicOr(&hdr->acc_flags, MGR_OVF_CUT);
                atomicOr(&hdr->rep_why, MGR_WHY_UNITS);
            }
        }
#ifdef MGR_TIMELINE
        if (lane == 0 && tlw_n < TLW_PER_WAVE) {
            const size_t k = ((size_t)blockIdx.x * 4 + wave) * TLW_PER_WAVE + tlw_n++;
            g_tlw[k * 4 + 0] = tlw0; g_tlw[k * 4 + 1] = wall_clock64();
            g_tlw[k * 4 + 2] = nlist | ((unsigned long long)quad << 28) | ((unsigned long long)blockIdx.x << 32); g_tlw[k * 4 + 3] = mx | (hole ? 1ull << 40 : 0ull);
        }
#endif
        FP(5);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef FWD_PROF
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&g_fprof[k], (unsigned long long)facc_[k]);
        atomicAdd(&g_fprof[6], (unsigned long long)fnt_);
        atomicAdd(&g_fprof[7], 1ull);
    }
#endif
    // empty tiles: background only (no per-pixel state: the backward has no work item that reads it)
    for (uint32_t q = n_busy + blockIdx.x; q < (uint32_t)VT; q += gridDim.x) {
        const uint32_t vt = tile_queue[q];
        const int v = (int)(vt / (uint32_t)T), t = (int)(vt % (uint32_t)T);
        const int px = (t % gx) * 16 + (tid & 15), py = (t / gx) * 16 + (tid >> 4);
        if (!skip_fill && px < W && py < H) {   // (skip_fill: the extra workgroups of k_dbin_sort have written the background)
            const size_t pix = (size_t)py * W + px;
            float* o = out_color + (size_t)v * 3 * P + pix;
            o[0] = bg0;
            o[P] = bg1;
            o[2 * P] = bg2;
        }
        if (tid == 0) tile_done[vt] = 0;
    }
}

// ---------------------------------------------------------------------------
// Depth cut, repaired on the device (mgr_common.h, MgrRepUnit).
// ---------------------------------------------------------------------------
// k_repair_prep: one workgroup per view collects the view's repaired tiles (the owner units of the view) into MgrRepView: tile,
// owner unit, depth of the cut, end of the depth window (tile_zwin) and the bounding box of the tiles -- so that the hundreds of
// workgroups of k_repair_scan start from one coalesced read instead of each walking the unit list and its dependent loads
// (unit -> tile -> hint: measured 0.045 ms per launch that way, most of it that latency).
__global__ __launch_bounds__(256) void k_repair_prep(int T, int gx, MgrHeader* hdr, const MgrRep rep, const uint32_t* __restrict__ tile_zused,
                                                     const uint32_t* __restrict__ tile_zwin) {
    const uint32_t nu = min(hdr->n_rep_units, rep.max_units);
    const int tid = threadIdx.x, v = blockIdx.x;
    MgrRepView* rv = rep.view + v;
    __shared__ uint32_t s_n, s_bb[4];
    if (tid == 0) { s_n = 0u; s_bb[0] = 0xFFFFu; s_bb[1] = 0xFFFFu; s_bb[2] = 0u; s_bb[3] = 0u; }
    __syncthreads();
    for (uint32_t u = (uint32_t)tid; u < nu; u += 256u) {
        const uint32_t vt = rep.unit[u].vt;
        if ((int)(vt / (uint32_t)T) == v && rep.tile_rep[vt] == u + 1u) {
            const uint32_t k = atomicAdd(&s_n, 1u);
            if (k < (uint32_t)MGR_REP_VIEW_TILES) {
                const uint32_t t = vt % (uint32_t)T, zcb = ~tile_zused[vt];      // float bits of the depth of the cut
                uint32_t zwb = tile_zwin[vt];
                if (zwb <= zcb) zwb = 0x7F7FFFFFu;                                // no usable window on file: everything behind the cut
                MgrRepTile e;
                e.tile = t; e.unit = u; e.zc = zcb; e.zw = zwb;
                rv->t[k] = e;
                const uint32_t ty = t / (uint32_t)gx, tx = t - ty * (uint32_t)gx;
                atomicMin(&s_bb[0], tx); atomicMin(&s_bb[1], ty); atomicMax(&s_bb[2], tx + 1u); atomicMax(&s_bb[3], ty + 1u);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        rv->n = s_n; rv->x0 = s_bb[0]; rv->y0 = s_bb[1]; rv->x1 = s_bb[2]; rv->y1 = s_bb[3];
        if (s_n > (uint32_t)MGR_REP_VIEW_TILES) { atomicOr(&hdr->acc_flags, MGR_OVF_CUT); atomicOr(&hdr->rep_why, MGR_WHY_VIEW_TILES); }
    }
}

// k_repair_scan: grid (segments of the instance range, views).  An instance the cut dropped from a repaired tile (rectangle of
// at most 64 tiles covering the tile, depth bits behind the cut, alive by the very cull test pre_tail applied: same inputs from
// the 48-byte record, same contraction-free functions) is a candidate = (depth bits << 32 | Gaussian) key of the tile's buffer,
// in whatever order -- k_repair_blend sorts.
//  * Depth window.  A repaired walk needs tens to hundreds of entries; everything behind a deep tile's cut is thousands (up to
//    21 000 on the bench scene).  Candidates behind the tile's window (tile_zwin: the depth of the entry MGR_REP_TARGET behind
//    the kept ones when the hint was made) are only counted (`beyond`); a walk that reaches the end of its window unsaturated
//    with instances behind it flags the forward (MGR_WHY_WINDOW: the legacy answer).
//  * The candidates of a workgroup are staged in LDS and handed to the tile's buffer with ONE global atomic per (workgroup,
//    tile): thousands of returning atomics on one address are served one after the other at ~11 ns each (measured: 0.157 ms
//    per launch with an atomic per candidate).
#define REP_SCAN_THREADS 256
#define REP_SCAN_SEGS 128
#define REP_STASH 1024
#define REP_HITS 3072
__global__ __launch_bounds__(REP_SCAN_THREADS) void k_repair_scan(int N, int T, int gx, MgrHeader* hdr, const MgrRep rep,
                                                                  const int32_t* __restrict__ radii, const ushort4* __restrict__ rect,
                                                                  const float* __restrict__ depth, const MgrGRec* __restrict__ grec) {
    if (min(hdr->n_rep_units, rep.max_units) == 0u) return;
    const int tid = threadIdx.x, v = blockIdx.y;
    const MgrRepView* rv = rep.view + v;
    const uint32_t nt = rv->n;
    if (nt == 0u || nt > (uint32_t)MGR_REP_VIEW_TILES) return;
    __shared__ uint32_t s_stash_n, s_nhit;
    __shared__ uint32_t s_tile[MGR_REP_VIEW_TILES], s_unit[MGR_REP_VIEW_TILES], s_zc[MGR_REP_VIEW_TILES], s_zw[MGR_REP_VIEW_TILES];
    __shared__ uint32_t s_hits[MGR_REP_VIEW_TILES], s_beyond[MGR_REP_VIEW_TILES];
    __shared__ uint4 s_stash[REP_STASH];   // (tile slot, rank among the workgroup's hits of that tile, depth bits, Gaussian)
    __shared__ uint2 s_hit[REP_HITS];      // (instance, tile slot | tile x << 8 | tile y << 20) of a batch
    __shared__ uint32_t s_bits[2048];      // one bit per tile of the view (the ordered binning serves grids of up to 65535 tiles)
    if (tid == 0) s_stash_n = 0u;
    for (int k = tid; k < 2048; k += REP_SCAN_THREADS) s_bits[k] = 0u;
    __syncthreads();
    for (uint32_t k = (uint32_t)tid; k < nt; k += REP_SCAN_THREADS) {
        const MgrRepTile e = rv->t[k];
        s_tile[k] = e.tile; s_unit[k] = e.unit; s_zc[k] = e.zc; s_zw[k] = e.zw; s_hits[k] = 0u; s_beyond[k] = 0u;
        atomicOr(&s_bits[e.tile >> 5], 1u << (e.tile & 31u));
    }
    const int bx0 = (int)rv->x0, by0 = (int)rv->y0, bx1 = (int)rv->x1, by1 = (int)rv->y1;
    __syncthreads();
    // Two phases per batch of instances.  (1) rectangle against the bitmap, LDS only: (instance, repaired tile) hits go to an LDS
    // list.  (2) the hits, one per thread: visibility, depth, the exact cull test -- the three global loads of a hit (radius,
    // depth, 48-byte record) are then in flight for 256 hits at once.  (With the loads inside the rectangle loop a wave paid
    // their round trip once per hit, lane by lane as the loop diverged: 0.079 ms per launch for six repaired tiles.)
    auto settle = [&](uint32_t i, uint32_t k, int x, int y) {
        const size_t vi = (size_t)v * N + i;
        if (radii[vi] <= 0) return;
        const uint32_t zbits = __float_as_uint(depth[vi]);
        if (zbits <= s_zc[k]) return;                            // in front of the cut: listed already
        const MgrGRec g = grec[vi];
        const MgrCull cull = mgr_cull_init(g.x, g.y, g.ca, g.cb, g.cc, mgr_qmax(g.op));
        float dy_lo, dy_hi, dxo;
        mgr_cull_row(cull, 16.0f * y, 16.0f * y + 15.0f, dy_lo, dy_hi, dxo);
        if (mgr_cull_dead(cull, dy_lo, dy_hi, dxo, 16.0f * x, 16.0f * x + 15.0f)) return;
        if (zbits > s_zw[k]) { atomicAdd(&s_beyond[k], 1u); return; }      // behind the depth window: counted only
        const uint32_t slot = atomicAdd(&s_stash_n, 1u);
        if (slot < (uint32_t)REP_STASH) {
            s_stash[slot] = make_uint4(k, atomicAdd(&s_hits[k], 1u), zbits, i);
        } else {      // (more candidates in one workgroup than the stage holds: straight to the buffer)
            const uint32_t u = s_unit[k], gslot = atomicAdd(&rep.cnt[u], 1u);
            if (gslot < (uint32_t)MGR_REP_CAND) rep.cand[(size_t)u * MGR_REP_CAND + gslot] = ((unsigned long long)zbits << 32) | i;
        }
    };
    const int per = (N + REP_SCAN_SEGS - 1) / REP_SCAN_SEGS, i_lo = (int)blockIdx.x * per, i_hi = min(N, i_lo + per);
    if (tid == 0) s_nhit = 0u;
    __syncthreads();
    // (one pass over the workgroup's whole segment, no barrier inside: with a barrier per batch of 1024 instances the five
    // batches of a segment each paid the rectangle loads' and the hits' round trips one after the other: 0.055 ms per launch)
    for (int ib = i_lo; ib < i_hi; ib += 6 * REP_SCAN_THREADS) {
        ushort4 rc[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) rc[r] = rect[(size_t)v * N + min(ib + r * REP_SCAN_THREADS + tid, N - 1)];      // six loads in flight
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int i = ib + r * REP_SCAN_THREADS + tid;
            const int x0 = rc[r].x, y0 = rc[r].y, x1 = rc[r].z, y1 = rc[r].w;
            const int tiles = (x1 - x0) * (y1 - y0);
            if (i >= i_hi || tiles <= 0 || tiles > 64) continue;            // (larger rectangles are never cut: pre_tail)
            if (x1 <= bx0 || x0 >= bx1 || y1 <= by0 || y0 >= by1) continue;
            for (int y = max(y0, by0); y < min(y1, by1); ++y)
                for (int x = max(x0, bx0); x < min(x1, bx1); ++x) {
                    const uint32_t t = (uint32_t)(y * gx + x);
                    if (!((s_bits[t >> 5] >> (t & 31u)) & 1u)) continue;
                    uint32_t k = 0;
                    while (k < nt && s_tile[k] != t) ++k;
                    if (k == nt) continue;
                    const uint32_t h = atomicAdd(&s_nhit, 1u);
                    if (h < (uint32_t)REP_HITS) s_hit[h] = make_uint2((uint32_t)i, k | ((uint32_t)x << 8) | ((uint32_t)y << 20));
                    else settle((uint32_t)i, k, x, y);                      // (list full: settled on the spot)
                }
        }
    }
    __syncthreads();
    {
        const uint32_t nh = min(s_nhit, (uint32_t)REP_HITS);
        for (uint32_t h = (uint32_t)tid; h < nh; h += REP_SCAN_THREADS) {
            const uint2 e = s_hit[h];
            settle(e.x, e.y & 0xFFu, (int)((e.y >> 8) & 0xFFFu), (int)(e.y >> 20));
        }
    }
    __syncthreads();
    for (uint32_t k = (uint32_t)tid; k < nt; k += REP_SCAN_THREADS) {
        const uint32_t c = s_hits[k], by = s_beyond[k], u = s_unit[k];
        s_hits[k] = c ? atomicAdd(&rep.cnt[u], c) : 0u;          // the workgroup's first slot in the tile's buffer
        if (by) atomicAdd(&rep.unit[u].beyond, by);
    }
    __syncthreads();
    const uint32_t ns = min(s_stash_n, (uint32_t)REP_STASH);
    for (uint32_t q = (uint32_t)tid; q < ns; q += REP_SCAN_THREADS) {
        const uint4 e = s_stash[q];
        const uint32_t u = s_unit[e.x], g = s_hits[e.x] + e.y;
        if (g < (uint32_t)MGR_REP_CAND) rep.cand[(size_t)u * MGR_REP_CAND + g] = ((unsigned long long)e.z << 32) | e.w;
    }
}

// k_repair_blend: workgroups of RS_THREADS threads take the owner units.  Per repaired tile: the candidates are sorted in LDS
// (lds_sort_emit: (depth, index) order = the tail of the tile's full list) and written behind the regular lists, after a copy
// of the cut list's last partial chunk; then waves 0..3 continue the walks of the tile's registered quadrants from their
// saved state through the sorted tail, with the forward blend's own pair step (mgr_fwd_pair_step), its box test and its
// checkpoint rule -- batches end on the chunk boundaries of the FULL list's positions, so the checkpoints are those the walk of
// the full list writes.  An empty tail needs nothing: the cut list was the full list.
#define REP_BLEND_LDS ((size_t)MGR_REP_CAND * 8 + RS_WAVES * 256 * 4 + 32 * 4 + 8 * 4 + 4 * 32 * MGR_PAIR_FLOATS * 4)
__global__ __launch_bounds__(RS_THREADS) void k_repair_blend(int N, int W, int H, int gx, int T, MgrHeader* hdr, const MgrRep rep,
                                                             const float* __restrict__ bg, uint32_t* __restrict__ sorted_gid,
                                                             const MgrGRec* __restrict__ grec, float* __restrict__ out_color,
                                                             uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_qdone,
                                                             float4* __restrict__ ckpt) {
    const uint32_t nu = min(hdr->n_rep_units, rep.max_units);
    if (nu == 0u) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned long long* s_keys = (unsigned long long*)s_raw;                       // MGR_REP_CAND keys
    uint32_t* s_cnt = (uint32_t*)(s_raw + (size_t)MGR_REP_CAND * 8);               // RS_WAVES x 256 counters
    uint32_t* s_scan = s_cnt + RS_WAVES * 256;                                     // 32 words
    uint32_t* s_q = s_scan + 32;                                                   // [4] unit + 1 of the tile's quadrants; [4..7] allocation
    float* s_slab = (float*)(s_q + 8);                                             // [4][32][MGR_PAIR_FLOATS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t P = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t u = blockIdx.x; u < nu; u += gridDim.x) {
        __syncthreads();
        const MgrRepUnit ru = rep.unit[u];
        if (rep.tile_rep[ru.vt] != u + 1u) continue;                               // not the tile's owner (workgroup-uniform)
        const uint32_t nraw = rep.cnt[u];
        if (nraw == 0u) {                                                          // nothing behind the cut: the cut list was the full list
            if (ru.beyond != 0u && tid == 0) {                                     // ... unless what there is lies behind the depth window
                atomicOr(&hdr->acc_flags, MGR_OVF_CUT);
                atomicOr(&hdr->rep_why, MGR_WHY_WINDOW);
            }
            continue;
        }
        if (tid < 4) s_q[tid] = 0u;
        __syncthreads();
        for (uint32_t k = (uint32_t)tid; k < nu; k += RS_THREADS) {
            const MgrRepUnit o = rep.unit[k];
            if (o.vt == ru.vt) s_q[o.quad & 3u] = k + 1u;
        }
        const uint32_t n = min(nraw, (uint32_t)MGR_REP_CAND);
        const uint32_t nl = ru.nlist, p0 = (nl / MGR_CHUNK) * MGR_CHUNK, pre = nl - p0, total = nl + n;
        const uint32_t c0 = p0 / MGR_CHUNK, c_last = (total - 1u) / MGR_CHUNK, nck = c_last - c0 + 1u;
        if (tid == 0) {
            const uint32_t ov = atomicAdd(&hdr->rep_list_used, pre + n), ck = atomicAdd(&hdr->rep_ck_used, nck);
            const bool fits = ov + pre + n <= rep.list_cap && ck + nck <= rep.ck_cap, ok = nraw <= (uint32_t)MGR_REP_CAND && fits;
            s_q[4] = ok ? 1u : 0u; s_q[5] = rep.list_base + ov; s_q[6] = rep.ck_base + ck;
            if (!ok) {                                                              // a capacity of the repair is exceeded: the legacy answer
                atomicOr(&hdr->acc_flags, MGR_OVF_CUT);
                atomicOr(&hdr->rep_why, fits ? MGR_WHY_CAND : MGR_WHY_LIST);
            }
        }
        __syncthreads();
        if (s_q[4] == 0u) continue;
        const uint32_t ov_start = s_q[5], ov_ck0 = s_q[6];
        // the tail in (depth, index) order: ids to the appended list, keys stay in LDS for the walks
        lds_sort_emit(rep.cand + (size_t)u * MGR_REP_CAND, n, s_keys, s_cnt, s_scan, tid, sorted_gid + ov_start + pre);
        if ((uint32_t)tid < pre) sorted_gid[ov_start + tid] = sorted_gid[ru.start + p0 + tid];
        if (tid == 0) {
            MgrRepUnit w = ru;
            w.ntail = n; w.ov_start = ov_start; w.ov_ck0 = ov_ck0; w.p0 = p0;
            w.ck_first = pre == 0u ? ov_ck0 : (p0 ? ru.ck0 + c0 - 1u : 0u);
            rep.unit[u] = w;
        }
        __syncthreads();
        if (wave >= 4 || s_q[wave] == 0u) continue;
        // ---- one wave per registered quadrant: the walk goes on ----
        const uint32_t uq = s_q[wave] - 1u;
        const MgrRepUnit rq = rep.unit[uq];
        const uint32_t vt = ru.vt, quad = rq.quad;
        const int v = (int)(vt / (uint32_t)T), t = (int)(vt % (uint32_t)T);
        const int bx = t % gx, by = t / gx;
        const int px = bx * 16 + (int)(quad & 1u) * 8 + (lane & 7), py = by * 16 + (int)(quad >> 1) * 8 + (lane >> 3);
        const bool inside = px < W && py < H;
        const mgr_v2f fpx2 = {(float)px, (float)px}, fpy2 = {(float)py, (float)py};
        const float qx0 = (float)(bx * 16 + (int)(quad & 1u) * 8), qy0 = (float)(by * 16 + (int)(quad >> 1) * 8);
        const int pslot = (int)(quad << 6) | lane;
        const float4 st = rep.state[(size_t)uq * 64 + lane];
        float Tr = st.w, C2 = st.z;
        mgr_v2f C01 = {st.x, st.y};
        uint32_t last = rep.last[(size_t)uq * 64 + lane];
        const unsigned long long exec_m = __builtin_amdgcn_ballot_w64(true);
        unsigned long long done_m = ((unsigned long long)rq.done_hi << 32) | rq.done_lo;
        const MgrGRec* const gv = grec + (size_t)v * N;
        float* const slab = s_slab + wave * 32 * MGR_PAIR_FLOATS;
        // the checkpoint in front of chunk c0 when the cut list ended exactly there (the forward blend writes a checkpoint only
        // in front of a chunk that exists in ITS list): the saved state is that checkpoint
        if (pre == 0u && p0 != 0u && (((~done_m & exec_m) >> lane) & 1ull)) ckpt[(size_t)ov_ck0 * 256 + pslot] = make_float4(C01.x, C01.y, C2, Tr);
        uint32_t j = 0;
        while (j < n) {
            const uint32_t blen = min(n - j, (uint32_t)MGR_CHUNK - ((nl + j) % MGR_CHUNK));   // up to the next chunk boundary of the full list
            int bx0, by0, bx1, by1;
            if (!mgr_quad_bbox(~done_m & exec_m, bx0, by0, bx1, by1)) break;
            bool alive = false;
            MgrGRec r;
            if ((uint32_t)lane < blen) {
                r = gv[(uint32_t)s_keys[j + lane]];
                alive = !mgr_box_dead(r.x, r.y, r.ca, r.cb, r.cc, mgr_qmax(r.op), qx0 + (float)bx0, qy0 + (float)by0, qx0 + (float)bx1,
                                      qy0 + (float)by1);
            }
            const unsigned long long m = __ballot(alive);
            const int cnt = __popcll(m);
            if (alive) {
                const int rank = __popcll(m & lt);
                float* pb = slab + (rank >> 1) * MGR_PAIR_FLOATS;
                mgr_pair_store<true>(pb, rank & 1, r.x, r.y, r.ca, r.cb, r.cc, r.op, r.r, r.g, r.b, nl + j + (uint32_t)lane + 1u);   // 1-based position in the FULL list
                if ((cnt & 1) && rank == cnt - 1) mgr_pair_pad<true>(pb);
            }
            __builtin_amdgcn_wave_barrier();
            const int npair = (cnt + 1) >> 1;
            for (int p = 0; p < npair; ++p) {
                const float4* pp = (const float4*)(slab + p * MGR_PAIR_FLOATS);
                const float4 R0 = pp[0], R1 = pp[1], R2 = pp[2], R3 = pp[3], R4 = pp[4];
                if (mgr_fwd_pair_step(R0, R1, R2, R3, R4, fpx2, fpy2, Tr, C01, C2, last, done_m, exec_m)) break;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t nextpos = nl + j + blen;
            if ((nextpos % MGR_CHUNK) == 0u && nextpos < total && (((~done_m & exec_m) >> lane) & 1ull))
                ckpt[(size_t)(ov_ck0 + nextpos / MGR_CHUNK - c0) * 256 + pslot] = make_float4(C01.x, C01.y, C2, Tr);
            j += blen;
        }
        if (inside) {
            const size_t pix = (size_t)py * W + px;
            n_contrib[(size_t)v * P + pix] = last;
            float* o = out_color + (size_t)v * 3 * P + pix;
            o[0] = C01.x + Tr * bg0;
            o[P] = C01.y + Tr * bg1;
            o[2 * P] = C2 + Tr * bg2;
        }
        const uint32_t mx = mgr_wave_max_u32(last);
        if (lane == 0) {
            tile_qdone[(size_t)vt * 4 + quad] = mx;      // (tile_qend keeps "never": no hint from a repaired walk)
            // the walk used up its depth window with a pixel unsaturated and instances behind the window: what lies there is needed
            if ((~done_m & exec_m) != 0ull && ru.beyond != 0u) { atomicOr(&hdr->acc_flags, MGR_OVF_CUT); atomicOr(&hdr->rep_why, MGR_WHY_WINDOW); }
        }
    }
}

// The backward blend's work items, built after the wave-granular forward blend: one 32-byte record per (tile, 64-entry
// chunk the tile consumed) = (tile, chunk, list offset of the chunk's first entry, checkpoint in front of the chunk | list
// depth consumed by each quadrant).  One thread per queue entry; a block reserves its range with one atomic and writes it
// with all its threads (record i belongs to the tile found by a search of the block's scan).  Also leaves the tile's depth
// in tile_done (the scheduling hint of the next forward).
//
// Depth cut: the same thread leaves the tile's hint for the NEXT forward in tile_zcut.  A tile whose every pixel had stopped
// by list position e (tile_qend) needs its entries up to e only; the hint keeps everything up to the depth
//   zc = max(z_e + range (z_e - z_0) + rel z_e,  z of entry e + max(min_entries, frac e))
// (z_0, z_e: depth of the first entry and of entry e - 1; the four margins: mgr_raster_set_cut_margin) and lets the next
// forward drop what lies behind: a margin in depth and in entries for whatever moved in between -- an optimizer step moves
// every opacity logit by its learning rate, so the walks of a deep tile lengthen by a percent or two of their length per
// step: the entry margin is proportional to the walk, and the engine widens all four when a forward is flagged.  (The bench scene packs ~270 list entries per millimetre of depth
// into its deep tiles: with 1/4 of the range and z / 500 the cut kept 56 % of the saturating tiles' pairs where the walks
// end after 18 %; tools/instr/cut_stats.py.)  A list already cut that ends inside that margin keeps at least its
// cut; an uncut one that does is needed whole.  Unsaturated tiles (silhouette, thin parts) get no hint.
// margins of the depth-cut hints (per host thread; see k_fwd_items)
static thread_local float g_cut_frac = 0.125f, g_cut_range = 0.0625f, g_cut_rel = 2.0e-4f;
static thread_local int g_cut_min = 64, g_cut_interior = 0, g_cut_penalty = 16;
extern "C" int mgr_raster_set_cut_penalty(int forwards) {
    if (forwards < 0 || forwards > 0x7FFFFFF) return mgr_fail(MGR_EINVAL, "mgr_raster_set_cut_penalty: 0 .. 2^27 forwards");
    g_cut_penalty = forwards;
    return MGR_OK;
}
extern "C" int mgr_raster_set_cut_margin(float frac_entries, int min_entries, float depth_range_frac, float depth_rel,
                                         int interior_only) {
    if (!(frac_entries >= 0.f) || min_entries < 0 || !(depth_range_frac >= 0.f) || !(depth_rel >= 0.f))
        return mgr_fail(MGR_EINVAL, "mgr_raster_set_cut_margin: margins must be non-negative");
    g_cut_frac = frac_entries; g_cut_min = min_entries; g_cut_range = depth_range_frac; g_cut_rel = depth_rel;
    g_cut_interior = interior_only ? 1 : 0;
    return MGR_OK;
}
__global__ __launch_bounds__(256) void k_fwd_items(const uint4* __restrict__ tile_qrec, const uint32_t* __restrict__ tile_qdone,
                                                   uint32_t* __restrict__ tile_done, uint4* __restrict__ items, MgrHeader* hdr,
                                                   int N, int T, const uint32_t* __restrict__ sorted_gid, const float* __restrict__ depth,
                                                   const uint32_t* __restrict__ tile_zused, const uint32_t* __restrict__ tile_qend,
                                                   uint32_t* __restrict__ tile_zcut, uint32_t* mirror, float cut_frac, uint32_t cut_min,
                                                   float cut_range, float cut_rel, int gx, int interior_only,
                                                   const uint32_t* __restrict__ tile_queue, int VT, unsigned char* __restrict__ tile_bgok,
                                                   const float* out_color, const float* __restrict__ bg, const MgrRep rep, uint32_t cut_penalty,
                                                   uint32_t* __restrict__ tile_zwin, const IlListArgs ll, int n_item_blocks) {
    __shared__ uint32_t s_scan[8];
    if ((int)blockIdx.x >= n_item_blocks) {      // the workgroups behind the items': the span list of the image loss (il_list.h), when attached
        const int lb = (int)blockIdx.x - n_item_blocks;
        il_list_mapped_block(ll, lb % ll.nbx, lb / ll.nbx, s_scan);
        return;
    }
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_run[257];
    __shared__ uint4 s_qr[256], s_qd[256], s_rp[256];
    const int tid = threadIdx.x;
    const uint32_t n_busy = hdr->queue_len_i;
    const uint32_t nb = (n_busy + 255u) / 256u;
    // the forward's last kernel publishes the depth-cut flags (tile scan + blend) and consumes them -- by the LAST workgroup of the
    // grid, which has no tiles of the queue when the frame has empty tiles (the system-scope fence of the status mirror then waits
    // beside the others' work instead of in front of workgroup 0's)
    if ((int)blockIdx.x == n_item_blocks - 1 && tid == 0) {
        const uint32_t f = hdr->acc_flags;
        uint32_t ovf = hdr->overflow;
        if (f) { ovf |= f; hdr->overflow = ovf; hdr->acc_flags = 0u; }
        if (mirror) {   // the caller's host-mapped status words (mgr_raster_set_status_mirror): no copy, no launch
            // (bits 16.. of the overflow word: quadrants of depth-cut tiles repaired on the device in this forward)
            mirror[0] = hdr->total_pairs; mirror[1] = ovf | (min(hdr->n_rep_units, 0xFFFFu) << 16); mirror[2] = hdr->tiers | (min(hdr->sort_big, 0xFFu) << 8) | (min(hdr->sort_near_large, 0xFFu) << 16) | (min(hdr->sort_large, 0x7Fu) << 24);
            __threadfence_system();
            mirror[3] = 1u;
        }
    }
    // "image kept": the image of this forward is complete behind this kernel -- its empty tiles hold the background (written
    // by the fill or left from the image before), the header names the image and the colour
    for (uint32_t q = hdr->queue_len + blockIdx.x * 256u + (uint32_t)tid; q < (uint32_t)VT; q += (uint32_t)n_item_blocks * 256u) tile_bgok[tile_queue[q]] = 1;
    if (blockIdx.x == 0 && tid == 0) {
        const unsigned long long owner = (unsigned long long)(uintptr_t)out_color;
        hdr->img_seq = hdr->fwd_seq;
        hdr->img_owner[0] = (uint32_t)owner; hdr->img_owner[1] = (uint32_t)(owner >> 32);
        hdr->img_bg[0] = __float_as_uint(bg[0]); hdr->img_bg[1] = __float_as_uint(bg[1]); hdr->img_bg[2] = __float_as_uint(bg[2]);
    }
    if (blockIdx.x >= nb) return;
    // strided over the queue (which is ordered by depth): every block gets its share of the deep tiles
    const uint32_t q = (uint32_t)tid * nb + blockIdx.x;
    uint4 qr = make_uint4(0, 0, 0, 0), qd = make_uint4(0, 0, 0, 0), rp = make_uint4(0, 0, 0, 0);
    uint32_t nch = 0;
    if (q < n_busy && tile_qrec[q].x != MGR_HOLE) {
        qr = tile_qrec[q];
        qd = *(const uint4*)(tile_qdone + (size_t)qr.x * 4);
        // a tile the repair kernels extended: its entries from p0 on live behind the regular lists (MgrRepUnit)
        if (rep.max_units) {
            const uint32_t ro = rep.tile_rep[qr.x];
            if (ro != 0u) {
                const MgrRepUnit ru = rep.unit[ro - 1u];
                if (ru.ntail != 0u) rp = make_uint4(ru.p0 | 0x80000000u, ru.ov_start, ru.ov_ck0, ru.ck_first);
            }
        }
        const uint32_t tmax = max(max(qd.x, qd.y), max(qd.z, qd.w));
        tile_done[qr.x] = tmax;
        nch = (tmax + MGR_CHUNK - 1) / MGR_CHUNK;
        const uint32_t e = tile_qend[qr.x], nl = qr.z, used = tile_zused[qr.x];
        uint32_t hint = 0u;
        const uint32_t em = e + max(cut_min, (uint32_t)(cut_frac * (float)e));   // entries kept: the walk's + the margin
        // interior_only: all eight neighbours saturated as well (an empty neighbour shows the background).  A tile at the
        // silhouette stops saturating when an edge moves by a fraction of a pixel -- no margin helps a pixel that sees
        // through -- and with the optimizer in the loop some such tile flagged nearly every step (measured: 253 of 255);
        // the engine asks for it once a forward has been flagged.
        bool interior = e != 0xFFFFFFFFu && e > 0u;
        if (interior && interior_only) {
            const uint32_t t = qr.x % (uint32_t)T, vbase = qr.x - t;
            const int tx = (int)(t % (uint32_t)gx), ty = (int)(t / (uint32_t)gx), gy = T / gx;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = tx + dx, y = ty + dy;
                    if ((dx | dy) == 0 || x < 0 || y < 0 || x >= gx || y >= gy) continue;
                    const uint32_t q2 = tile_qend[vbase + (uint32_t)(y * gx + x)];
                    interior = interior && q2 != 0xFFFFFFFFu && q2 != 0u;
                }
        }
        // a countdown left by an earlier forward (k_tile_scan_b kept it): no hint while it runs; a tile whose cut list ran out
        // in THIS forward (repaired on the device, or flagged) starts one -- the same few tiles at the silhouette of the
        // saturating region otherwise run out step after step under a moving model
        const uint32_t pen = tile_zcut[qr.x];
        const bool ran_out = used != 0u && e == 0xFFFFFFFFu;
        if (ran_out) hint = cut_penalty;
        else if (pen > 1u) hint = pen - 1u;
        else if (interior && e <= nl && (em <= nl || used != 0u)) {
            const uint32_t* sg = sorted_gid + qr.y;
            const float* dv = depth + (size_t)(qr.x / (uint32_t)T) * N;
            const uint32_t g0 = sg[0], ge = sg[e - 1u], gm = sg[min(em, nl) - 1u];
            const float z0 = dv[g0], ze = dv[ge], zm = dv[gm];
            float zc = ze + cut_range * (ze - z0) + cut_rel * ze;
            // a list already cut that ends inside the margin has no entry to read the margin's depth from: extrapolate it
            // from the walk's own depth per entry (x 1.5), and never move the cut forward
            const float zx = fmaxf(__uint_as_float(~used), ze + 1.5f * (float)(em - e) * (ze - z0) / (float)e);
            zc = fmaxf(zc, em <= nl ? zm : zx);
            hint = ~__float_as_uint(zc);
            // the repair's depth window (k_repair_scan): a forward that walked the tile's FULL list knows the depth of the entry
            // MGR_REP_TARGET behind the kept ones (or that the list ends before: no window); a cut list does not reach that far:
            // the window on file stays (the cut creeps by a margin's fraction per step, the window is thousands of entries deep)
            if (used == 0u) {
                const uint32_t iw = max(em, e) + (uint32_t)MGR_REP_TARGET;
                tile_zwin[qr.x] = iw >= nl ? 0x7F7FFFFFu : __float_as_uint(fmaxf(dv[sg[iw - 1u]], zc));
            }
        }
        tile_zcut[qr.x] = hint;
    }
    uint32_t total;
    const uint32_t run = block_excl_scan(nch, s_scan, total);
    s_run[tid] = run;
    s_qr[tid] = qr;
    s_qd[tid] = qd;
    s_rp[tid] = rp;
    if (tid == 0) {
        s_run[256] = total;
        s_base = total ? atomicAdd(&hdr->n_items, total) : 0u;
    }
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t i = (uint32_t)tid; i < total; i += 256u) {   // consecutive threads write consecutive records
        int lo = 0, hi = 256;   // the last tile whose first record is at or in front of i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_run[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t c = i - s_run[lo];
        const uint4 r = s_qr[lo], x = s_rp[lo];
        uint32_t off = r.y + c * MGR_CHUNK, ck = r.w + (c > 0 ? c - 1 : 0);
        const uint32_t p0 = x.x & 0x7FFFFFFFu;
        if (x.x != 0u && c * MGR_CHUNK >= p0) {      // chunk of a repaired tile at or behind the end of its cut list
            off = x.y + (c * MGR_CHUNK - p0);
            ck = c * MGR_CHUNK == p0 ? x.w : x.z + (c - p0 / MGR_CHUNK);
        }
        items[2 * (size_t)(base + i)] = make_uint4(r.x, c, off, ck);
        items[2 * (size_t)(base + i) + 1] = s_qd[lo];
    }
}

// ---------------------------------------------------------------------------
// host entry
// ---------------------------------------------------------------------------
struct CanonInputs {  // canonical (un-posed) parameters of the fused articulated path
    int B, n_art, sh_half;
    const float *xyz, *log_scale, *rot, *op_logit, *f_dc, *f_rest, *skin_w, *transforms;
};

// One side stream (+ fork/join events) per host thread for the kernels that can overlap.
struct MgrSideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipError_t init() {
        if (stream) return hipSuccess;
        hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        return hipEventCreateWithFlags(&join, hipEventDisableTiming);
    }
};
#define MGR_MAX_DEVICES 64
static MgrSideStream& mgr_side_stream(int device) {   // one per (host thread, device): streams and events belong to a device
    static thread_local MgrSideStream s[MGR_MAX_DEVICES];
    return s[device];
}

// Host-mapped status words per workspace (mgr_raster_set_status_mirror): the forward's last kernel writes (pair total,
// overflow word, binning tiers, 1) there, so that a caller who does not want to synchronise needs neither a device-to-host
// copy nor a read-back launch per forward -- an event behind the forward is enough.  One shot: taken by the forward that
// runs the blend on that workspace.
static std::mutex g_mirror_mu;
static struct { const void* ws; uint32_t* dev; } g_mirror[64];
static uint32_t* mgr_take_status_mirror(const void* workspace) {
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    for (auto& m : g_mirror)
        if (m.ws == workspace) { uint32_t* p = m.dev; m.ws = nullptr; m.dev = nullptr; return p; }
    return nullptr;
}
// The span list of the image loss, attached to the next forward that runs its last kernel on `workspace` (one shot, like the
// status mirror): k_fwd_items' launch then carries the list's workgroups (il_list.h) -- the list needs the forward's tile offsets
// only, and as a launch of its own it was 8 us of the chain of small kernels between the forward blend and the loss.
static struct { const void* ws; IlListArgs a; } g_loss_list[64];
static IlListArgs mgr_take_loss_list(const void* workspace) {
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    for (auto& m : g_loss_list)
        if (m.ws == workspace) { IlListArgs a = m.a; m.ws = nullptr; return a; }
    IlListArgs none;
    memset(&none, 0, sizeof(none));
    return none;
}
extern "C" int mgr_views_forward_attach_loss_list(const void* workspace, int V, int H, int W, const uint32_t* target_map,
                                                  void* loss_workspace, size_t loss_workspace_bytes) {
    if (!workspace) return mgr_fail(MGR_EINVAL, "mgr_views_forward_attach_loss_list: null workspace");
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    int free_slot = -1;
    for (int k = 0; k < 64; ++k) {
        if (g_loss_list[k].ws == workspace) { free_slot = k; break; }
        if (!g_loss_list[k].ws && free_slot < 0) free_slot = k;
    }
    if (!target_map) {      // withdraw
        if (free_slot >= 0 && g_loss_list[free_slot].ws == workspace) g_loss_list[free_slot].ws = nullptr;
        return MGR_OK;
    }
    if (V <= 0 || H <= 0 || W <= 0 || W > ILS_MAXW || !loss_workspace) return mgr_fail(MGR_EINVAL, "mgr_views_forward_attach_loss_list: bad arguments");
    if (loss_workspace_bytes < (size_t)il_blocks(V, H, W) * 12 + 256) return mgr_fail(MGR_ENOMEM, "mgr_views_forward_attach_loss_list: loss workspace too small");
    if (free_slot < 0) return mgr_fail(MGR_ENOMEM, "mgr_views_forward_attach_loss_list: too many pending lists");
    g_loss_list[free_slot].ws = workspace;
    g_loss_list[free_slot].a = il_list_args(V, H, W, target_map, nullptr, loss_workspace);
    return MGR_OK;
}

extern "C" int mgr_raster_set_status_mirror(const void* workspace, void* host_words) {
    if (!workspace) return mgr_fail(MGR_EINVAL, "mgr_raster_set_status_mirror: null workspace");
    void* dev = nullptr;
    if (host_words) MGR_HIP(hipHostGetDevicePointer(&dev, host_words, 0));
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    int free_slot = -1;
    for (int k = 0; k < 64; ++k) {
        if (g_mirror[k].ws == workspace) { free_slot = k; break; }
        if (!g_mirror[k].ws && free_slot < 0) free_slot = k;
    }
    if (free_slot < 0) return mgr_fail(MGR_EINVAL, "mgr_raster_set_status_mirror: too many workspaces with a pending mirror");
    g_mirror[free_slot].ws = host_words ? workspace : nullptr;
    g_mirror[free_slot].dev = (uint32_t*)dev;
    return MGR_OK;
}

static int raster_forward_impl(int V, int N, int W, int H, const float* cams, const float* bg,
                               const float* means3D, int64_t s_means, const float* cov3D,
                               int64_t s_cov, const float* colors, int64_t s_col,
                               const float* opacity, int64_t s_op, const CanonInputs* canon, float* out_color,
                               int32_t* radii, void* workspace, size_t workspace_bytes,
                               int64_t cap, int debug, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // debug bit 0: synchronise and check after every kernel; bit 1: stop before the blend (instance kernels and binning
    // only); bit 2: the blend only (after a call with bit 1 on the same workspace and arguments)
    // bit 3 (8): depth cut -- apply the hints the previous forward on this workspace left in tile_zcut (fused path only)
    // bits 4 / 5 (16 / 32): skip the binning launches for tile boxes of more than 2048 / of 1537..2048 tiles (the caller saw
    // in the previous forward's header that no view needed them; a view that does now raises MGR_OVF_TIER)
    const bool do_bin = !(debug & 4), do_blend = !(debug & 2), use_cut = (debug & 8) && canon != nullptr;
    static const bool bg_fill_on = [] { const char* e = getenv("MANUS_BG_FILL"); return !(e && e[0] == 'b'); }();   // MANUS_BG_FILL=blend: by the blend, as before (A/B)
    bool bg_filled = false;   // the background of the empty tiles has been written by the instance sort's launch
    const int skip_tiers = ((debug & 16) ? 1 : 0) | ((debug & 32) ? 2 : 0) | ((debug & 128) ? 4 : 0) | ((debug & 256) ? 8 : 0);
    const int img_kept = (debug & 1024) ? 1 : 0;   // bit 10: "image kept" (see BgFill)
    const bool spread = (debug & 4096) != 0;       // bit 12: k_bin_scatter's lane-spreading instantiation (the previous forward met rectangles of more than 64 tiles)
    // bit 11 (2048, with bit 3): tiles whose cut list runs out under an unsaturated pixel are repaired on the device
    // (k_repair_scan / k_repair_blend) instead of flagging the forward
    const bool repair = use_cut && (debug & 2048) && W < 65536 && H < 65536;   // (k_repair_scan packs tile coordinates in 12 bits)
    debug &= 1;
    if (V <= 0 || N < 0 || W <= 0 || H <= 0 || cap < 0 || cap > 0xFFFFFFF0ll)
        return mgr_fail(MGR_EINVAL, "mgr_raster_forward: bad sizes");
    if (!cams || !bg || !out_color || !workspace ||
        (N > 0 && !canon && (!means3D || !cov3D || !colors || !opacity)) || (N > 0 && !radii))
        return mgr_fail(MGR_EINVAL, "mgr_raster_forward: null pointer");
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    if (gx > 65535 || gy > 65535) return mgr_fail(MGR_EINVAL, "mgr_raster_forward: image too large");
    const MgrLayout L = mgr_layout(V, N, W, H, cap);
    if (workspace_bytes < L.total) return mgr_fail(MGR_ENOMEM, "mgr_raster_forward: workspace too small");
    char* ws = (char*)workspace;
    MgrHeader* hdr = (MgrHeader*)(ws + L.header);
    uint32_t* tile_count = (uint32_t*)(ws + L.tile_count);
    uint32_t* tile_start = (uint32_t*)(ws + L.tile_start);
    const int VT = V * T;
    MgrRep rep_all;
    rep_all.unit = (MgrRepUnit*)(ws + L.rep_unit); rep_all.state = (float4*)(ws + L.rep_state); rep_all.last = (uint32_t*)(ws + L.rep_last);
    rep_all.cnt = (uint32_t*)(ws + L.rep_cnt); rep_all.tile_rep = (uint32_t*)(ws + L.tile_rep); rep_all.cand = (unsigned long long*)(ws + L.rep_cand);
    rep_all.view = (MgrRepView*)(ws + L.rep_view);
    rep_all.max_units = repair ? mgr_rep_units(cap) : 0u;
    rep_all.list_cap = mgr_rep_list(cap); rep_all.ck_cap = mgr_rep_ck(cap);
    rep_all.list_base = (uint32_t)cap; rep_all.ck_base = (uint32_t)(cap / MGR_CHUNK + 1);

    // function attributes are per device: set once for each device a forward runs on (first call on that device)
    static std::atomic<bool> attr_set[MGR_MAX_DEVICES];
    int device = 0;
    MGR_HIP(hipGetDevice(&device));
    if (device < 0 || device >= MGR_MAX_DEVICES) return mgr_fail(MGR_EINVAL, "mgr_raster_forward: device index out of range");
    if (!attr_set[device].load(std::memory_order_acquire)) {
        MGR_HIP(hipFuncSetAttribute((const void*)k_tile_sort, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SORT_LDS_KEYS * 8 + RS_WAVES * 256 * 4 + 256));
        MGR_HIP(hipFuncSetAttribute((const void*)k_preprocess, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_emit, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_inst_fwd<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_inst_fwd<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_inst_fwd<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_inst_fwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_dbin_sort<SORT_LDS_KEYS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SORT_LDS_KEYS * 8 + RS_WAVES * 256 * 4 + 256));
        MGR_HIP(hipFuncSetAttribute((const void*)k_bin_count<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_bin_scatter<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_bin_scatter<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        MGR_HIP(hipFuncSetAttribute((const void*)k_repair_blend, hipFuncAttributeMaxDynamicSharedMemorySize, REP_BLEND_LDS));
        attr_set[device].store(true, std::memory_order_release);
    }

    const int lds_hist = ((size_t)T * 4 + 128 <= 150 * 1024) ? 1 : 0;
    // depth-ordered binning unless the tile grid does not fit the LDS cursors (or MGR_BINNING=sorted asks for the per-tile sorts)
    const char* binning_env = getenv("MGR_BINNING");   // read per call: tests flip it between two forwards
    const bool ordered_env = !(binning_env && strcmp(binning_env, "sorted") == 0);
    const bool ordered = ordered_env && lds_hist && T <= 65535 && (size_t)T * 4 + BIN_SC_FIXED_BYTES <= 150 * 1024;
    const size_t hist_bytes = lds_hist ? (size_t)T * 4 : 0;
    if (do_bin) {
    // (no per-call memset: the pair and flag accumulators of the header are consumed -- left zero -- by the kernels that
    // publish them, k_tile_scan_b and k_fwd_items; 7 us per forward as a launch of its own)
    // tile_count and the size-class counters are left zero by the previous forward on this workspace
    // (k_tile_scan_b / k_blend_fwd) and by the zero-filled allocation before the first one

    if (N > 0) {
        dim3 grid((N + PRE_THREADS - 1) / PRE_THREADS, V);
        if (canon) {
            MGR_PROF("k_inst_fwd", stream);
            const bool mixed = canon->skin_w && canon->n_art < N;
#define MGR_IF_LAUNCH(MX, HF)                                                                                           \
    hipLaunchKernelGGL((k_inst_fwd<MX, HF>), dim3(8 * ((grid.x + 7) / 8) * V), dim3(PRE_THREADS), 128 + hist_bytes, stream, N, canon->B, canon->n_art, W, H, \
                       gx, gy, cams, canon->xyz, canon->log_scale, canon->rot, canon->op_logit, canon->f_dc,            \
                       canon->f_rest, canon->skin_w, canon->transforms, (MgrGRec*)(ws + L.grec),                        \
                       (float*)(ws + L.depth), (ushort4*)(ws + L.rect), (unsigned long long*)(ws + L.alive),            \
                       (uint32_t*)(ws + L.pair_off), tile_count, radii, hdr, lds_hist, V,                                  \
                       use_cut ? (const uint32_t*)(ws + L.tile_zcut) : (const uint32_t*)nullptr, (uint32_t*)(ws + L.db_zrange))
            if (mixed && canon->sh_half) MGR_IF_LAUNCH(true, true);
            else if (mixed) MGR_IF_LAUNCH(true, false);
            else if (canon->sh_half) MGR_IF_LAUNCH(false, true);
            else MGR_IF_LAUNCH(false, false);
#undef MGR_IF_LAUNCH
        } else
        { MGR_PROF("k_preprocess", stream); hipLaunchKernelGGL(k_preprocess, grid, dim3(PRE_THREADS), 128 + hist_bytes, stream, N, W, H, gx,
                           gy, cams, means3D, s_means, cov3D, s_cov, colors, s_col, opacity, s_op,
                           (MgrGRec*)(ws + L.grec), (float*)(ws + L.depth), (ushort4*)(ws + L.rect),
                           (unsigned long long*)(ws + L.alive), (uint32_t*)(ws + L.pair_off), tile_count, radii,
                           hdr, lds_hist, (uint32_t*)(ws + L.db_zrange)); }
        MGR_LAUNCH_CHECK("k_preprocess", stream, debug);
    }
    {
        const int nbT = (T + 1023) / 1024, nblk = V * nbT;   // blocks of up to 1024 tiles of one view
        uint2* part = (uint2*)(ws + L.scan_part);
        uint32_t* blk_cls = (uint32_t*)(ws + L.scan_cls);
        const int use_hint = ordered ? 1 : 0;
        // ordered binning: the depth-bucket count rides behind the workgroups of scan phase A, the bucket scan behind those
        // of phase B (dbin_count_block / dbin_scan_block)
        const bool dbin = ordered && N > 0;
        const int n_bx = (N + 1024 * DB_PER - 1) / (1024 * DB_PER);
        uint4* blk_box = (uint4*)(ws + L.scan_box);
        const DbinArgs dba = {N, n_bx, (const int32_t*)radii, (const float*)(ws + L.depth), (const ushort4*)(ws + L.rect),
                              (const unsigned long long*)(ws + L.alive), dbin ? (uint32_t*)(ws + L.db_count) : nullptr,
                              (uint32_t*)(ws + L.db_cursor), (uint32_t*)(ws + L.db_start), (uint32_t*)(ws + L.db_nvis),
                              (ushort4*)(ws + L.db_bbox), (unsigned long long*)(ws + L.db_keys), (uint32_t*)(ws + L.db_zrange),
                              (uint32_t*)(ws + L.db_item), (N + MGR_DB_ITEM - 1) / MGR_DB_ITEM + 1, (N + PRE_THREADS - 1) / PRE_THREADS};
        { MGR_PROF("k_tile_scan_a", stream); hipLaunchKernelGGL(k_tile_scan_a, dim3(nblk + (dbin ? n_bx * V : 0)), dim3(1024), 0, stream, T, nbT, tile_count,
                           (const uint32_t*)(ws + L.tile_done), use_hint, part, blk_cls, blk_box, gx, nblk, dba, hdr); }
        { MGR_PROF("k_tile_scan_b", stream); hipLaunchKernelGGL(k_tile_scan_b, dim3(nblk + (dbin ? V : 0)), dim3(1024), 0, stream, V, T, nbT, tile_count, part, (const uint32_t*)blk_cls,
                           tile_start, (uint32_t*)(ws + L.tile_cursor), (uint32_t*)(ws + L.tile_queue), (uint4*)(ws + L.tile_qrec),
                           (const uint32_t*)(ws + L.tile_done), use_hint, (uint32_t*)(ws + L.chunk_start), hdr, (uint32_t)cap,
                           (uint32_t*)(ws + L.tile_zcut), (uint32_t*)(ws + L.tile_zused), (uint32_t*)(ws + L.tile_qend), use_cut ? 1 : 0, (const uint4*)blk_box, dba, (unsigned char*)(ws + L.tile_bgok), (uint32_t*)(ws + L.tile_rep), rep_all); }
    }
    MGR_LAUNCH_CHECK("k_tile_scan", stream, debug);
    if (N > 0 && ordered) {
        const int bb = mgr_bin_block(V, N), nblk = (N + bb - 1) / bb;
        const ushort4* rect = (const ushort4*)(ws + L.rect);
        const unsigned long long* alive = (const unsigned long long*)(ws + L.alive);
        uint32_t* db_start = (uint32_t*)(ws + L.db_start);
        uint32_t* db_nvis = (uint32_t*)(ws + L.db_nvis);
        unsigned long long* db_keys = (unsigned long long*)(ws + L.db_keys);
        uint32_t* db_order = (uint32_t*)(ws + L.db_order);
        uint32_t* bin_mat = (uint32_t*)(ws + L.bin_mat);
        uint4* db_rec = (uint4*)(ws + L.db_rec);
        ushort4* db_bbox = (ushort4*)(ws + L.db_bbox);
        const dim3 grid_n((N + 1024 * DB_PER - 1) / (1024 * DB_PER), V), grid_b(nblk, V);
        const int chunk = bb == MGR_BIN_BLOCK ? DB_CHUNK : 1024, chunks = (N + chunk - 1) / chunk;
        const size_t rec_bytes = BIN_SC_FIXED_BYTES;
        { MGR_PROF("k_dbin_scatter", stream); hipLaunchKernelGGL(k_dbin_scatter, grid_n, dim3(1024), 0, stream, N, (const int32_t*)radii, (const float*)(ws + L.depth), rect, alive,
                           (const uint32_t*)db_start, (uint32_t*)(ws + L.db_cursor), db_keys, (const uint32_t*)(ws + L.db_zrange), (N + PRE_THREADS - 1) / PRE_THREADS); }
        { MGR_PROF("k_dbin_sort", stream);
          // items of at most MGR_DB_RANK_MAX keys by counting, one workgroup each (k_dbin_rank); behind it the radix launch for
          // larger ones (returns at once when there are none) -- it also carries the background of the empty tiles (BgFill)
          const int ipv = (N + MGR_DB_ITEM - 1) / MGR_DB_ITEM + 1;
          const bool ranked = ipv <= DBR_MAX_ITEMS;      // (more instances per view than the scan can table: everything through the radix launch)
          // debug bit 128 (skip_tiers & 4): the previous forward met no item beyond k_dbin_rank -- the launch behind it is skipped
          // (7 us of workgroups that return at once); an item that needs it after all raises MGR_OVF_TIER
          const bool behind = !ranked || !(skip_tiers & 4);
          BgFill fill = {do_blend && bg_fill_on ? out_color : nullptr, (const uint32_t*)(ws + L.tile_queue), bg, VT, T, gx, W, H,
                         (const unsigned char*)(ws + L.tile_bgok), img_kept};
          const BgFill none = {nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0};
          const int n_fill = fill.out ? 1024 : 0;
          bg_filled = fill.out != nullptr;
          // debug bit 256 (skip_tiers & 8): the previous forward met items beyond MGR_DB_RANK_MAX keys (one dense depth bucket: the palm
          // seen face on, in the trained state of the bench scene) -- this launch gets LDS for MGR_DB_RANK_LARGE keys per workgroup
          const uint32_t cap_keys = (skip_tiers & 8) ? (uint32_t)MGR_DB_RANK_LARGE : (uint32_t)MGR_DB_RANK_MAX;
          if (ranked && (skip_tiers & 8))
              hipLaunchKernelGGL((k_dbin_rank<MGR_DB_RANK_LARGE>), dim3((unsigned)(ipv * V + n_fill)), dim3(DBR_THREADS), 0, stream, N, V, ipv, (const uint32_t*)db_start, (const uint32_t*)(ws + L.db_item),
                                 (const unsigned long long*)db_keys, db_order, hdr, behind ? 0 : 1, fill, ipv * V);
          else if (ranked)
              hipLaunchKernelGGL((k_dbin_rank<MGR_DB_RANK_MAX>), dim3((unsigned)(ipv * V + n_fill)), dim3(DBR_THREADS), 0, stream, N, V, ipv, (const uint32_t*)db_start, (const uint32_t*)(ws + L.db_item),
                                 (const unsigned long long*)db_keys, db_order, hdr, behind ? 0 : 1, fill, ipv * V);
          const int fchunk = ranked ? (int)MGR_DB_ITEM : chunk, fchunks = (N + fchunk - 1) / fchunk;
          (void)chunks;
          if (behind)
              hipLaunchKernelGGL((k_dbin_sort<SORT_LDS_KEYS>), dim3(1024 + (ranked ? 0 : n_fill)), dim3(RS_THREADS), SORT_LDS_KEYS * 8 + RS_WAVES * 256 * 4 + 256, stream,
                                 N, fchunk, fchunks, V * fchunks, (const uint32_t*)db_start, (const uint32_t*)db_nvis, db_keys, db_order,
                                 ranked ? cap_keys : 0u, hdr, ranked ? 2 : 0, ranked ? none : fill, 1024); }
        MGR_LAUNCH_CHECK("k_dbin_sort", stream, debug);
        const bool big_possible = T > BIN_SMALL_TILES;   // a box of more than BIN_SMALL_TILES tiles can only exist then
        { MGR_PROF("k_bin_count", stream);
          const dim3 grid_c = (V % 8 == 0) ? dim3(nblk * V) : grid_b;   // (XCD-aware order, see the kernel)
          hipLaunchKernelGGL((k_bin_count<true>), grid_c, dim3(BCNT_THREADS), (size_t)BIN_SMALL_TILES * 4, stream, N, T, nblk, bb, (const uint32_t*)db_nvis,
                             (const ushort4*)db_bbox, (const uint32_t*)db_order, rect, alive, db_rec, bin_mat, V % 8 == 0 ? 1 : 0);
          if (big_possible && !(skip_tiers & 1))
              hipLaunchKernelGGL((k_bin_count<false>), grid_c, dim3(BCNT_THREADS), (size_t)T * 4, stream, N, T, nblk, bb, (const uint32_t*)db_nvis,
                                 (const ushort4*)db_bbox, (const uint32_t*)db_order, rect, alive, db_rec, bin_mat, V % 8 == 0 ? 1 : 0); }
        { MGR_PROF("k_bin_scan", stream); hipLaunchKernelGGL(k_bin_scan, dim3((T + BSCAN_COLS - 1) / BSCAN_COLS, V), dim3(BSCAN_COLS * BSCAN_SEGS), 0, stream, gx, T, nblk, bb, (const uint32_t*)db_nvis,
                           (const ushort4*)db_bbox, (const uint32_t*)tile_start, bin_mat); }
        { MGR_PROF("k_bin_scatter", stream);
#define MGR_SC_LAUNCH(MK, LDSB, ...)                                                                                                          \
    do {                                                                                                                                  \
        if (spread) hipLaunchKernelGGL((k_bin_scatter<MK, true>), grid_b, dim3(BIN_SC_THREADS), LDSB, stream, __VA_ARGS__);             \
        else hipLaunchKernelGGL((k_bin_scatter<MK, false>), grid_b, dim3(BIN_SC_THREADS), LDSB, stream, __VA_ARGS__);                   \
    } while (0)
          MGR_SC_LAUNCH(BIN_MID_TILES, (size_t)BIN_MID_TILES * 12 + rec_bytes, N, T, nblk, bb,
                             (const uint32_t*)db_nvis, (const ushort4*)db_bbox, (const uint32_t*)db_order, (const uint4*)db_rec, (const uint32_t*)bin_mat,
                             (uint32_t*)(ws + L.sorted_gid), (uint32_t)cap, hdr, skip_tiers);
          if (T > BIN_MID_TILES && !(skip_tiers & 2))
              MGR_SC_LAUNCH(BIN_SMALL_TILES, (size_t)BIN_SMALL_TILES * 12 + rec_bytes, N, T, nblk, bb,
                            (const uint32_t*)db_nvis, (const ushort4*)db_bbox, (const uint32_t*)db_order, (const uint4*)db_rec, (const uint32_t*)bin_mat,
                            (uint32_t*)(ws + L.sorted_gid), (uint32_t)cap, hdr, 0);
          if (big_possible && !(skip_tiers & 1))
              MGR_SC_LAUNCH(0, (size_t)T * 4 + rec_bytes, N, T, nblk, bb,
                            (const uint32_t*)db_nvis, (const ushort4*)db_bbox, (const uint32_t*)db_order, (const uint4*)db_rec, (const uint32_t*)bin_mat,
                            (uint32_t*)(ws + L.sorted_gid), (uint32_t)cap, hdr, 0);
#undef MGR_SC_LAUNCH
        }
        MGR_LAUNCH_CHECK("k_bin_scatter", stream, debug);
    } else if (N > 0) {
        dim3 grid((N + EMIT_THREADS - 1) / EMIT_THREADS, V);
        { MGR_PROF("k_emit", stream); hipLaunchKernelGGL(k_emit, grid, dim3(EMIT_THREADS), hist_bytes + 16, stream, N, gx, gy,
                           (const float*)(ws + L.depth), (const ushort4*)(ws + L.rect),
                           (const unsigned long long*)(ws + L.alive), tile_start,
                           (uint32_t*)(ws + L.tile_cursor), (unsigned long long*)(ws + L.keys),
                           (uint32_t)cap, lds_hist); }
        MGR_LAUNCH_CHECK("k_emit", stream, debug);
        // The small-tile sort only depends on k_emit; it runs on a side stream next to the
        // giant-tile split (few workgroups, long) and the LDS radix sort (one workgroup per CU),
        // and joins before the blend.
        MgrSideStream& side = mgr_side_stream(device);
        MGR_HIP(side.init());
        MGR_HIP(hipEventRecord(side.fork, stream));
        MGR_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
        // (the split is launched first so that its few 1024-thread workgroups are placed before the small-tile sort
        // fills the CUs: launched after it, they waited for room and the kernel took three times its own work)
        { MGR_PROF("k_tile_split", stream); hipLaunchKernelGGL(k_tile_split, dim3(256), dim3(SORT_THREADS), 0, stream,
                           tile_start, (const uint32_t*)(ws + L.tile_queue), (unsigned long long*)(ws + L.keys),
                           (unsigned long long*)(ws + L.keys2), (uint4*)(ws + L.groups),
                           (uint32_t*)(ws + L.sorted_gid), hdr, (uint32_t)cap); }
        { MGR_PROF("k_tile_sort_small", side.stream); hipLaunchKernelGGL(k_tile_sort_small, dim3(256 * 8), dim3(256), 0, side.stream,
                           tile_start, (const uint32_t*)(ws + L.tile_queue),
                           (const unsigned long long*)(ws + L.keys), (uint32_t*)(ws + L.sorted_gid), hdr,
                           (uint32_t)cap); }
        MGR_HIP(hipEventRecord(side.join, side.stream));
        { MGR_PROF("k_tile_sort", stream); hipLaunchKernelGGL(k_tile_sort, dim3(512), dim3(RS_THREADS), SORT_LDS_KEYS * 8 + RS_WAVES * 256 * 4 + 256, stream,
                           tile_start, (const uint32_t*)(ws + L.tile_queue),
                           (const unsigned long long*)(ws + L.keys), (const unsigned long long*)(ws + L.keys2),
                           (const uint4*)(ws + L.groups), (uint32_t*)(ws + L.sorted_gid), hdr, (uint32_t)cap); }
        MGR_HIP(hipStreamWaitEvent(stream, side.join, 0));
        MGR_LAUNCH_CHECK("k_tile_sort", stream, debug);
    }
    }   // do_bin
    if (!do_blend) return MGR_OK;
    { MGR_PROF("k_blend_fwd", stream); hipLaunchKernelGGL(k_blend_fwd, dim3(MGR_FWD_GRID), dim3(256), 0, stream, N, W, H, gx, gy, VT, bg,
                       (const uint32_t*)(ws + L.tile_queue), (const uint4*)(ws + L.tile_qrec), (const uint32_t*)(ws + L.sorted_gid), (const MgrGRec*)(ws + L.grec), out_color,
                       (uint32_t*)(ws + L.n_contrib), (uint32_t*)(ws + L.tile_done), (uint32_t*)(ws + L.tile_qdone),
                       (float4*)(ws + L.ckpt), hdr, (const uint32_t*)(ws + L.tile_zused), (uint32_t*)(ws + L.tile_qend), bg_filled ? 1 : 0); }
    if (repair && N > 0) {   // (both return at once when the blend registered no unit)
        { MGR_PROF("k_repair_scan", stream);
          hipLaunchKernelGGL(k_repair_prep, dim3(V), dim3(256), 0, stream, T, gx, hdr, rep_all, (const uint32_t*)(ws + L.tile_zused),
                             (const uint32_t*)(ws + L.tile_zwin));
          hipLaunchKernelGGL(k_repair_scan, dim3(REP_SCAN_SEGS, V), dim3(REP_SCAN_THREADS), 0, stream, N, T, gx, hdr, rep_all, (const int32_t*)radii,
                             (const ushort4*)(ws + L.rect), (const float*)(ws + L.depth), (const MgrGRec*)(ws + L.grec)); }
        { MGR_PROF("k_repair_blend", stream);
          hipLaunchKernelGGL(k_repair_blend, dim3(128), dim3(RS_THREADS), REP_BLEND_LDS, stream, N, W, H, gx, T, hdr, rep_all, bg,
                             (uint32_t*)(ws + L.sorted_gid), (const MgrGRec*)(ws + L.grec), out_color, (uint32_t*)(ws + L.n_contrib),
                             (uint32_t*)(ws + L.tile_qdone), (float4*)(ws + L.ckpt)); }
        MGR_LAUNCH_CHECK("k_repair_blend", stream, debug);
    }
    // the span list of the image loss rides on this launch when the caller attached one (mgr_views_forward_attach_loss_list)
    IlListArgs ll = mgr_take_loss_list(workspace);
    if (ll.nbx) {
        if (ll.H != H || ll.W != W || ll.list_views != V) return mgr_fail(MGR_EINVAL, "attached loss list: views or image size differ from the forward's");
        ll.tile_start = tile_start;
    }
    const int n_item_blocks = (VT + 255) / 256;
    { MGR_PROF("k_fwd_items", stream); hipLaunchKernelGGL(k_fwd_items, dim3((unsigned)(n_item_blocks + ll.nbx * V)), dim3(256), 0, stream, (const uint4*)(ws + L.tile_qrec),
                       (const uint32_t*)(ws + L.tile_qdone), (uint32_t*)(ws + L.tile_done), (uint4*)(ws + L.items), hdr,
                       N, T, (const uint32_t*)(ws + L.sorted_gid), (const float*)(ws + L.depth), (const uint32_t*)(ws + L.tile_zused),
                       (const uint32_t*)(ws + L.tile_qend), (uint32_t*)(ws + L.tile_zcut), mgr_take_status_mirror(workspace),
                       g_cut_frac, (uint32_t)g_cut_min, g_cut_range, g_cut_rel, gx, g_cut_interior,
                       (const uint32_t*)(ws + L.tile_queue), VT, (unsigned char*)(ws + L.tile_bgok), (const float*)out_color, bg, rep_all,
                       (uint32_t)g_cut_penalty, (uint32_t*)(ws + L.tile_zwin), ll, n_item_blocks); }
    MGR_LAUNCH_CHECK("k_blend_fwd", stream, debug);
    return MGR_OK;
}

extern "C" int mgr_raster_forward(int V, int N, int W, int H, const float* cams, const float* bg,
                                  const float* means3D, int64_t s_means, const float* cov3D,
                                  int64_t s_cov, const float* colors, int64_t s_col,
                                  const float* opacity, int64_t s_op, float* out_color,
                                  int32_t* radii, void* workspace, size_t workspace_bytes,
                                  int64_t cap, int debug, void* stream_) {
    return raster_forward_impl(V, N, W, H, cams, bg, means3D, s_means, cov3D, s_cov, colors, s_col, opacity, s_op,
                               nullptr, out_color, radii, workspace, workspace_bytes, cap, debug, stream_);
}

extern "C" int mgr_views_forward(int V, int N, int B, int n_articulated, int sh_half, int W, int H, const float* cams, const float* bg,
                                 const float* xyz, const float* log_scale, const float* rot,
                                 const float* opacity_logit, const float* f_dc, const float* f_rest,
                                 const float* skin_w, const float* transforms, float* out_color,
                                 int32_t* radii, void* workspace, size_t workspace_bytes, int64_t cap,
                                 int debug, void* stream_) {
    if (N > 0 && (!xyz || !log_scale || !rot || !opacity_logit || !f_dc || !f_rest || (skin_w && !transforms)))
        return mgr_fail(MGR_EINVAL, "mgr_views_forward: null pointer");
    if (skin_w && (B <= 0 || B > MGR_MAX_BONES)) return mgr_fail(MGR_EINVAL, "mgr_views_forward: bad B");
    if (skin_w && (n_articulated < 0 || n_articulated > N)) return mgr_fail(MGR_EINVAL, "mgr_views_forward: bad n_articulated");
    if (sh_half && ((uintptr_t)f_rest & 15)) return mgr_fail(MGR_EINVAL, "mgr_views_forward: fp16 f_rest must be 16-byte aligned");
    const CanonInputs ci = {B, skin_w ? n_articulated : 0, sh_half ? 1 : 0, xyz, log_scale, rot, opacity_logit, f_dc, f_rest, skin_w, transforms};
    return raster_forward_impl(V, N, W, H, cams, bg, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, &ci, out_color,
                               radii, workspace, workspace_bytes, cap, debug, stream_);
}

#ifdef BIN_PROF
extern "C" int mgr_binprof(unsigned long long* out) {
    MGR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_binprof), 8 * 8 * 4096));
    return 0;
}
#endif
extern "C" int mgr_raster_layout(int V, int N, int W, int H, int64_t cap, size_t* out, int n_out) {
    const MgrLayout L = mgr_layout(V, N, W, H, cap);
    const size_t v[] = {L.header, L.grec, L.depth, L.rect, L.alive, L.pair_off, L.tile_count, L.tile_start,
                        L.tile_cursor, L.tile_done, L.tile_queue, L.chunk_start, L.items, L.ckpt, L.keys,
                        L.sorted_gid, L.final_T, L.n_contrib, L.pair_tag, L.pair_grad, L.total, L.inst_grad, L.inst_tag,
                        L.db_nvis, L.db_bbox, L.db_order, L.tile_zcut, L.tile_zused, L.tile_qend, L.tile_rep, L.rep_unit, L.rep_cnt, L.tile_zwin};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < n_out; ++i) out[i] = v[i];
    return n;
}

__global__ void k_debug_pair_alpha(int n, const float* __restrict__ rec, const int32_t* __restrict__ px,
                                   const int32_t* __restrict__ py, float* __restrict__ alpha, int32_t* __restrict__ valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float pb[MGR_PAIR_FLOATS];
    const float* r = rec + (size_t)i * 6;
    mgr_pair_store(pb, 0, r[0], r[1], r[2], r[3], r[4], r[5], 0.f, 0.f, 0.f, 1u);   // the staging the blend kernels use
    mgr_pair_pad(pb);
    const float4 R0 = make_float4(pb[0], pb[1], pb[2], pb[3]), R1 = make_float4(pb[4], pb[5], pb[6], pb[7]),
                 R2 = make_float4(pb[8], pb[9], pb[10], pb[11]);
    const mgr_v2f fpx2 = {(float)px[i], (float)px[i]}, fpy2 = {(float)py[i], (float)py[i]};
    mgr_v2f dx, dy, G, al;
    bool va, vb;
    mgr_pair_alpha(R0, R1, R2, fpx2, fpy2, dx, dy, G, al, va, vb);
    alpha[i] = al.x;
    valid[i] = va ? 1 : 0;
}

extern "C" int mgr_debug_pair_alpha(int n, const float* rec, const int32_t* px, const int32_t* py, float* alpha,
                                    int32_t* valid, void* stream_) {
    if (n < 0 || (n > 0 && (!rec || !px || !py || !alpha || !valid))) return mgr_fail(MGR_EINVAL, "mgr_debug_pair_alpha: bad arguments");
    if (n == 0) return MGR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_debug_pair_alpha, dim3((n + 255) / 256), dim3(256), 0, stream, n, rec, px, py, alpha, valid);
    MGR_LAUNCH_CHECK("k_debug_pair_alpha", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_raster_status_sync(const void* workspace, int64_t* num_pairs, int32_t* overflow,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    uint32_t h[2] = {0, 0};
    MGR_HIP(hipMemcpyAsync(h, workspace, 8, hipMemcpyDeviceToHost, stream));
    MGR_HIP(hipStreamSynchronize(stream));
    if (num_pairs) *num_pairs = h[0];
    if (overflow) *overflow = (int32_t)h[1];
    if (h[1] & MGR_OVF_PAIRS) return mgr_fail(MGR_EOVERFLOW, "pair capacity exceeded");
    if (h[1] & MGR_OVF_CUT) return mgr_fail(MGR_ECUT, "depth cut violated: run the forward again without debug bit 8");
    if (h[1] & MGR_OVF_TIER) return mgr_fail(MGR_ETIER, "a skipped binning tier was needed: run the forward again without debug bits 16 / 32");
    return MGR_OK;
}

extern "C" int mgr_raster_status_tiers_sync(const void* workspace, int64_t* num_pairs, int32_t* overflow, int32_t* tiers,
                                            void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    uint32_t h[64];      // (the first 256 bytes of the header: everything but the queue counters)
    static_assert(offsetof(MgrHeader, sort_near_large) / 4 < 64, "status words beyond the copied part of the header");
    MGR_HIP(hipMemcpyAsync(h, workspace, sizeof(h), hipMemcpyDeviceToHost, stream));
    MGR_HIP(hipStreamSynchronize(stream));
    if (num_pairs) *num_pairs = h[0];
    if (overflow) *overflow = (int32_t)h[1];
    if (tiers) {   // bits 0-1: binning tiers needed; bits 8..15 / 16..23: items of the instance sort beyond 13/16 of
                   // MGR_DB_RANK_MAX / MGR_DB_RANK_LARGE keys (capped at 255); bits 24..30: items beyond MGR_DB_RANK_MAX keys (the
                   // next forward should ask for the large instantiation: debug bit 256)
        const uint32_t sb = h[offsetof(MgrHeader, sort_big) / 4], sn = h[offsetof(MgrHeader, sort_near_large) / 4], sl = h[offsetof(MgrHeader, sort_large) / 4];
        *tiers = (int32_t)(h[offsetof(MgrHeader, tiers) / 4] | ((sb < 0xFFu ? sb : 0xFFu) << 8) | ((sn < 0xFFu ? sn : 0xFFu) << 16) | ((sl < 0x7Fu ? sl : 0x7Fu) << 24));
    }
    if (h[1] & MGR_OVF_PAIRS) return mgr_fail(MGR_EOVERFLOW, "pair capacity exceeded");
    if (h[1] & MGR_OVF_CUT) return mgr_fail(MGR_ECUT, "depth cut violated: run the forward again without debug bit 8");
    if (h[1] & MGR_OVF_TIER) return mgr_fail(MGR_ETIER, "a skipped binning tier was needed: run the forward again without debug bits 16 / 32");
    return MGR_OK;
}

extern "C" int mgr_raster_debug_binning_sync(const void* workspace, int V, int N, int W, int H,
                                             int64_t cap, int view, int32_t* tile_ranges_host,
                                             int32_t* point_list_host, int64_t max_pairs,
                                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    if (view < 0 || view >= V) return mgr_fail(MGR_EINVAL, "bad view");
    const MgrLayout L = mgr_layout(V, N, W, H, cap);
    const char* ws = (const char*)workspace;
    uint32_t* ts = (uint32_t*)malloc((size_t)(T + 1) * 4);
    MGR_HIP(hipMemcpyAsync(ts, ws + L.tile_start + (size_t)view * T * 4, (size_t)(T + 1) * 4,
                           hipMemcpyDeviceToHost, stream));
    MGR_HIP(hipStreamSynchronize(stream));
    const uint32_t b = ts[0], e = ts[T];
    for (int t = 0; t < T; ++t) {
        tile_ranges_host[2 * t] = (int32_t)(ts[t] - b);
        tile_ranges_host[2 * t + 1] = (int32_t)(ts[t + 1] - b);
    }
    free(ts);
    int64_t n = (int64_t)e - (int64_t)b;
    if (n > max_pairs) n = max_pairs;
    if (n > 0 && point_list_host) {
        MGR_HIP(hipMemcpyAsync(point_list_host, ws + L.sorted_gid + (size_t)b * 4, (size_t)n * 4,
                               hipMemcpyDeviceToHost, stream));
        MGR_HIP(hipStreamSynchronize(stream));
    }
    return MGR_OK;
}
