// Shared device/host helpers for libmanus_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/manus_hip.h"

#define MGR_WAVE 64

// ---------------------------------------------------------------------------
// error handling: thread-local text, never abort()
// ---------------------------------------------------------------------------
extern thread_local char g_mgr_err[512];

static inline int mgr_fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_mgr_err, sizeof(g_mgr_err), fmt, a, b);
    return code;
}

#define MGR_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return mgr_fail(MGR_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// after a launch: catch launch errors always; in debug mode also sync + check
#define MGR_LAUNCH_CHECK(name, stream, debug)                                              \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) return mgr_fail(MGR_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
        if (debug) {                                                                       \
            e_ = hipStreamSynchronize(stream);                                             \
            if (e_ != hipSuccess) return mgr_fail(MGR_EHIP, "sync %s: %s", name, hipGetErrorString(e_)); \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------
// optional per-kernel timing with HIP events on the caller's stream
// (mgr_profile_enable / mgr_profile_report); zero cost when disabled
// ---------------------------------------------------------------------------
void mgr_prof_begin(const char* name, hipStream_t stream);
void mgr_prof_end(hipStream_t stream);
extern int g_mgr_prof_on;
bool mgr_prof_match(const char* name);   // mgr_profile_filter: only the named kernel (empty = all of them)
struct MgrProfScope {
    hipStream_t s;
    bool on;
    MgrProfScope(const char* name, hipStream_t stream) : s(stream), on(g_mgr_prof_on != 0 && mgr_prof_match(name)) {
        if (on) mgr_prof_begin(name, s);
    }
    ~MgrProfScope() {
        if (on) mgr_prof_end(s);
    }
};
#define MGR_PROF(name, stream) MgrProfScope mgr_prof_scope_(name, stream)

// ---------------------------------------------------------------------------
// workspace layout of the rasterizer (shared by forward and backward)
// ---------------------------------------------------------------------------
// repair of depth-cut tiles on the device (described below, next to MGR_CHUNK)
#define MGR_WHY_UNITS 1u        // more quadrants ran out than the repair has units for (or the repair is off)
#define MGR_WHY_VIEW_TILES 2u   // more repaired tiles in one view than MGR_REP_VIEW_TILES
#define MGR_WHY_CAND 4u         // more instances behind a tile's cut (inside the depth window) than MGR_REP_CAND
#define MGR_WHY_LIST 8u         // the appended entries / checkpoints are used up
#define MGR_WHY_EMPTY 16u       // a hinted tile ended up with no list at all
#define MGR_WHY_WINDOW 32u      // a repaired walk reached the end of its depth window unsaturated with instances left behind it
#define MGR_REP_CAND 8192          // instances behind the cut one repaired tile may hold (one LDS sort)
#define MGR_REP_TARGET 3072        // ... of which the scan aims for this many: the depth window behind the cut (tile_zwin, k_fwd_items)
#define MGR_REP_VIEW_TILES 256     // repaired tiles per view and forward
struct __attribute__((aligned(16))) MgrRepUnit {      // 64 bytes
    uint32_t vt, quad, nlist, start;      // tile (view * T + tile), quadrant, length and offset of the CUT list
    uint32_t ck0, done_lo, done_hi, ntail;   // first checkpoint of the tile; lanes whose walk had ended; entries appended (owner unit; 0: none)
    uint32_t ov_start, ov_ck0, p0, ck_first;   // owner unit: list offset of entry p0 = 64 * (nlist / 64), checkpoint in front of chunk c > p0 / 64 at ov_ck0 + c - p0 / 64, of chunk p0 / 64 at ck_first
    uint32_t beyond;                      // owner unit: instances behind the cut AND behind the depth window (not collected)
    uint32_t pad[3];
};
struct __attribute__((aligned(16))) MgrRepTile {      // one repaired tile of a view (k_repair_prep -> k_repair_scan)
    uint32_t tile, unit, zc, zw;          // tile of the view, owner unit, float bits of the depth of the cut / of the end of the depth window
};
struct MgrRepView {       // per view: its repaired tiles of the forward in flight
    uint32_t n, x0, y0, x1, y1, pad[3];   // count; bounding box of the repaired tiles (tile units, x1 / y1 exclusive)
    MgrRepTile t[MGR_REP_VIEW_TILES];
};
struct MgrRep {           // device pointers of the repair state (max_units == 0: repair off)
    MgrRepUnit* unit;
    float4* state;        // [unit][64] prefix colour + transmittance of the quadrant's pixels at the end of the cut list
    uint32_t* last;       // [unit][64] last contributor
    uint32_t* cnt;        // [unit] candidates found (owner units)
    uint32_t* tile_rep;   // [V * T] owner unit + 1 of a repaired tile (0: none)
    unsigned long long* cand;   // [unit][MGR_REP_CAND] (depth bits << 32 | Gaussian)
    MgrRepView* view;     // [V]
    uint32_t max_units, list_cap, ck_cap, list_base, ck_base;   // capacities; first appended entry / checkpoint index
};
struct MgrHeader {            // first 256 bytes of the workspace
    uint32_t total_pairs;     // sum of tiles_touched over all views (may exceed capacity); published by k_tile_scan_b
    uint32_t overflow;        // MGR_OVF_* bits of the most recent forward (k_tile_scan_b: pairs; k_fwd_items adds the depth-cut flag)
    uint32_t epoch;           // backward call counter (tags valid pair-gradient records)
    uint32_t queue_len;       // number of non-empty tiles in tile_queue
    uint32_t queue_head;      // work-queue cursor (sort)
    uint32_t queue_head2;     // spare cursor
    uint32_t n_items;         // number of (tile, chunk) work items appended by the forward blend
    uint32_t item_head;       // backward work-queue cursor
    uint32_t queue_small;     // queue index of the first tile with fewer than 2048 pairs
    uint32_t queue_head3;     // cursor of the small-tile sort
    uint32_t n_active;        // Gaussians with a non-zero pair gradient in the current view group (fused backward)
    uint32_t queue_len_i;     // length of the view-interleaved queue of the forward blend (tile_qrec, with holes)
    // accumulators of the forward in flight: pairs counted by the per-instance kernel, depth-cut flags raised by the tile
    // scan and the blend.  Whoever publishes them (above) leaves them zero for the next forward -- no memset per call.
    uint32_t acc_pairs, acc_flags;
    // ordered binning: LDS tiers the views' tile boxes of the most recent forward needed beyond the smallest one
    // (bit 0: a box of more than 2048 tiles, bit 1: one of 1537..2048) -- a caller may skip the launches of tiers the
    // previous forward did not need (debug bits 16 / 32); a skipped tier that IS needed raises MGR_OVF_TIER
    uint32_t tiers;
    uint32_t sort_big;        // items of the instance sort holding a single bucket of more than DBS_LIGHT_KEYS keys (bits 8.. of the reported tiers word)
    // fused backward, 5..8 views per group: the active Gaussians by the number of their views that hold pair records, rounded up
    // to 8 / 4 / 2 lanes -- k_inst_bwd_runs gives a Gaussian that many lanes instead of eight (k_inst_gather counts and lists them)
    uint32_t n_runs[4];       // (three classes; the fourth word is spare)
    // "outputs kept" (mgr_views_backward, debug bit 512): backward calls on this workspace (k_blend_bwd counts), the call whose
    // gather left the per-Gaussian row state (which gradient rows of the caller's buffers may be non-zero), the gather's note
    // for the kernel behind it
    uint32_t bwd_seq, rows_seq, rows_pending;
    uint32_t rows_owner[2];   // d_xyz of the call that left the row state: the state describes THOSE buffers
    // "image kept" (forward, debug bit 1024): forwards binned on this workspace (k_tile_scan_a counts), the forward that last
    // completed an image (k_fwd_items), the image it wrote and its background colour -- tile_bgok says which tiles of THAT image
    // hold the background
    uint32_t fwd_seq, img_seq, img_owner[2], img_bg[3];
    // depth cut, repaired on the device (forward debug bit 2048; see MgrRepUnit): (tile, quadrant) units registered by the blend
    // of the forward in flight, entries / checkpoints handed out behind the regular lists; reset by k_tile_scan_b
    uint32_t n_rep_units, rep_list_used, rep_ck_used;
    uint32_t rep_why;         // why the forward in flight raised MGR_OVF_CUT (bits MGR_WHY_*; k_tile_scan_a clears it)
    MgrRep rep;               // the repair's pointers / capacities of the forward in flight (k_tile_scan_b copies its argument here: the
                              // forward blend reads them in its rare "list ran out" branch instead of carrying 17 more scalars)
    uint32_t sort_huge;       // items of the instance sort beyond the LDS of the k_dbin_rank launch (it counts; the launch behind it returns at once on 0)
    uint32_t sort_large;      // items beyond MGR_DB_RANK_MAX keys, whatever the launch's LDS (bits 24.. of the reported tiers word: the caller asks
                              // the next forward for the large LDS, debug bit 256)
    uint32_t sort_near_large; // items beyond 13/16 of MGR_DB_RANK_LARGE (bits 16..23 of the reported tiers word; sort_big -- bits 8..15 --
                              // counts those beyond 13/16 of MGR_DB_RANK_MAX: the caller skips the launch behind by the one that
                              // belongs to the instantiation it is going to ask for)
    uint32_t spare[45 - sizeof(MgrRep) / 4];
    uint32_t queue_giant;     // queue index of the first tile with fewer than 16384 pairs
    uint32_t n_groups;        // depth groups produced by k_tile_split for the giant tiles
    uint32_t split_head, group_head;
    uint32_t pad2[56 - 8 - 68 + 64 - 4];
    // work-queue counters of the backward blend, one per 256-byte line: same-address global atomics are served one
    // after the other (~11 ns each on this chip, device-wide: 79 000 tickets on one counter were measured to take
    // 0.9 ms), so a queue that hands out tens of thousands of wave-sized items is spread over MGR_NCTR addresses
    uint32_t qctr[16 * 64];
    // the same for the forward blend, whose work units are (tile, 8x8 quadrant) = one wave each; k_tile_scan_b presets
    // them so that the first ticket of every counter lies behind the units the waves start on statically
    uint32_t qctr_f[16 * 64];
};
#define MGR_NCTR 16
// bits of MgrHeader::overflow
#define MGR_OVF_PAIRS 1u   // the pair capacity was exceeded: lists clipped, image and gradients incomplete
#define MGR_OVF_CUT 2u     // a tile whose list was cut short by the depth cut ran out of entries with a pixel still unsaturated
#define MGR_OVF_TIER 4u    // a view's tile box needed a binning launch the caller had asked to skip (debug bits 16 / 32)

// Per-(view, Gaussian) record gathered by the blend kernels: 48 bytes of content in a 64-byte slot on a 64-byte boundary.
// The memory side serves gathers in 128-byte requests (profiles/r06_counter_calibration.txt): a quarter of 48-byte records at
// 16-byte alignment straddle two of them, a 64-byte slot never does -- k_blend_bwd 0.350 -> 0.344 ms, k_blend_fwd 0.239 -> 0.238,
// step 1.332 -> 1.318 ms (round 6 A/B, one box; -DMGR_GREC_BYTES=48 builds the packed layout again).
#ifndef MGR_GREC_BYTES
#define MGR_GREC_BYTES 64
#endif
struct __attribute__((aligned(MGR_GREC_BYTES == 64 ? 64 : 16))) MgrGRec {
    float x, y, ca, cb;       // pixel centre, conic A, B
    float cc, op, r, g;       // conic C, opacity, colour r, g
    float b;                  // colour b
    int32_t slot_base;        // pair slot = slot_base + ty*rect_w + tx
    int32_t rect_w;
    int32_t pad;
#if MGR_GREC_BYTES == 64
    int32_t pad64[4];
#endif
};

#define MGR_CHUNK 64         // list entries per backward work item / forward checkpoint interval (one batch of the blend waves)

// ---------------------------------------------------------------------------
// Depth cut, repaired on the device (round 6; mgr_views_forward debug bits 8 + 2048).
//
// A forward with the depth cut drops, per tile, the instances behind the depth at which the tile's pixels had all saturated
// in the previous forward (+ a margin).  When the model moves, a few tiles per step run out of list under a pixel that has
// not saturated (measured with Adam at the reference's learning rates on the bench scene: 6 tiles per step at the default
// margins, 36 at half of them -- tools/instr/cut_flag_stats.py).  Up to round 5 such a forward was flagged and the WHOLE
// step run again on full lists, which made the cut useless under an optimizer.  Now the blend wave that runs out registers
// a repair unit = (tile, quadrant, the 64 pixels' state); behind the blend
//   k_repair_scan   finds, per repaired tile, the instances the cut dropped from it (one pass over the view's rectangles:
//                   rectangle covers the tile, depth behind the cut, not null by the exact cull test of the binning),
//   k_repair_blend  sorts them by (depth, index) = the tail of the tile's full list, appends it behind the regular lists
//                   (the last partial 64-entry chunk of the cut list is copied in front of it, so that the backward's
//                   chunks stay contiguous), and walks it from the saved state: image, n_contrib, checkpoints, consumed
//                   depth -- bit for bit what the walk of the full list leaves,
// and k_fwd_items points the backward's work items of the repaired tiles at the appended lists / checkpoints.  The forward
// is only flagged (MGR_OVF_CUT: re-run without the cut) when a capacity below is exceeded.
// ---------------------------------------------------------------------------
// repair capacities as a function of the pair capacity (small scenes: small workspaces)
static inline uint32_t mgr_rep_units(int64_t cap) { int64_t u = cap / 16384; return (uint32_t)(u < 64 ? 64 : (u > 1024 ? 1024 : u)); }
static inline uint32_t mgr_rep_list(int64_t cap) { int64_t n = cap / 8; return (uint32_t)(n < (1 << 16) ? (1 << 16) : (n > (2 << 20) ? (2 << 20) : n)); }
static inline uint32_t mgr_rep_ck(int64_t cap) { return mgr_rep_list(cap) / MGR_CHUNK + mgr_rep_units(cap); }

struct MgrLayout {
    size_t header, scan_part, scan_cls, scan_box, grec, depth, rect, alive, pair_off, tile_count, tile_start, tile_cursor, tile_done, tile_qdone,
        tile_zcut, tile_zused, tile_qend, tile_bgok, tile_queue, tile_qrec, chunk_start, items, ckpt, keys, keys2, groups, sorted_gid, final_T, n_contrib, pair_tag, pair_grad, inst_grad, inst_tag,
        db_count, db_cursor, db_start, db_nvis, db_bbox, db_keys, db_order, db_rec, bin_mat, tile_rep, rep_unit, rep_state, rep_last, rep_cnt, rep_cand, rep_view, tile_zwin, db_zrange, db_item, total;
};

// Depth-ordered binning (raster_fwd.hip, "ordered" route): the instances of a view are sorted by depth once, then
// scattered to the tile lists in that order, instead of sorting every tile list.
#ifndef MGR_DB_BUCKETS
#define MGR_DB_BUCKETS 1024   // depth buckets per view of the instance sort (uniform over the depth range of the view's visible instances).
                              // The densest depth slice of the bench hand (the palm seen face on) is a bucket of 1569 keys, 2196 after 600 Adam
                              // steps (tools/instr/depth_bucket_stats.py): that item goes to the launch behind k_dbin_rank.  More buckets keep
                              // every item within k_dbin_rank but cost the bucket scatter one global add per non-empty bucket and block: same
                              // box, eight views, 2048 buckets 1.312-1.322 ms against 1.308-1.311 (with the optimizer 1.507 against 1.488), 4096
                              // buckets 1.324 against 1.315 -- not adopted.
#endif
#ifndef MGR_DB_ITEM
#define MGR_DB_ITEM 768       // keys per item of the instance sort (whole buckets: an item ends with the bucket it is in)
#endif
#define MGR_DB_RANK_MAX 2048  // items of at most this many keys are sorted by k_dbin_rank (16 + 8 KB of LDS), larger ones by the launch behind it --
                              // 33 us on the critical path -- unless the caller asked for the instantiation for MGR_DB_RANK_LARGE keys (debug bit
                              // 256: the previous forward met such items), which costs k_dbin_rank ~4 us (six keys per thread compiled in) and
                              // spares the launch behind.  (3584: eight keys per thread, +15 us.)
#define MGR_DB_RANK_LARGE 3072
#ifndef MGR_BIN_BLOCK
#define MGR_BIN_BLOCK 1024
#endif
// MGR_BIN_BLOCK:    // depth-consecutive instances per row of the (block, tile) count matrix (at most)
// rows of 256 instances when there are few instances in all (one or two views per rank): k_bin_scatter runs one wave per row
static inline int mgr_bin_block(int V, int N) { return (long long)V * (long long)N >= 500000ll ? MGR_BIN_BLOCK : 256; }

static inline size_t mgr_align(size_t x) { return (x + 255) & ~(size_t)255; }

static inline MgrLayout mgr_layout(int V, int N, int W, int H, int64_t cap) {
    MgrLayout L;
    size_t VN = (size_t)V * (size_t)(N > 0 ? N : 1);
    size_t gx = (W + MGR_TILE - 1) / MGR_TILE, gy = (H + MGR_TILE - 1) / MGR_TILE;
    size_t VT = (size_t)V * gx * gy, VP = (size_t)V * W * H;
    size_t c = (size_t)(cap > 0 ? cap : 1);
    size_t o = 0;
    L.header = o;      o += mgr_align(sizeof(MgrHeader));
    const size_t scan_blocks = (size_t)V * ((gx * gy + 1023) / 1024);      // the tile scan runs blocks of up to 1024 tiles of one view
    L.scan_part = o;   o += mgr_align((scan_blocks + 1) * 8);         // per-block (pairs, chunks) sums of the tile scan
    L.scan_cls = o;    o += mgr_align(scan_blocks * 34 * 4);          // per-block histogram of its tiles over the size classes
    L.scan_box = o;    o += mgr_align(scan_blocks * 16);              // per-block bounding box of its non-empty tiles (x0, y0, x1, y1)
    L.grec = o;        o += mgr_align(VN * sizeof(MgrGRec));
    L.depth = o;       o += mgr_align(VN * 4);
    L.rect = o;        o += mgr_align(VN * 8);       // 4 x uint16
    L.alive = o;       o += mgr_align(VN * 8);       // bitmask of non-null tiles of the rect
    L.pair_off = o;    o += mgr_align(VN * 4);
    L.tile_count = o;  o += mgr_align(VT * 4);
    L.tile_start = o;  o += mgr_align((VT + 1) * 4);
    L.tile_cursor = o; o += mgr_align(VT * 4);
    L.tile_done = o;   o += mgr_align(VT * 4);
    L.tile_qdone = o;  o += mgr_align(VT * 16);       // list depth each 8x8 quadrant of a tile consumed in the forward blend
    // depth cut (mgr_views_forward, debug bit 8): per tile the complement of the float bits of the depth beyond which the
    // NEXT forward may leave instances out of the tile's list (0 = no cut), written by k_fwd_items from how deep THIS
    // forward's walk went; the value a forward applied (k_tile_scan_b moves it there); the list position at which the
    // tile's last quadrant saturated (0xFFFFFFFF: some pixel never did)
    L.tile_zcut = o;   o += mgr_align(VT * 4);
    L.tile_zused = o;  o += mgr_align(VT * 4);
    L.tile_qend = o;   o += mgr_align(VT * 4);
    L.tile_queue = o;  o += mgr_align(VT * 4);
    L.tile_qrec = o;   o += mgr_align(VT * 16);       // queue of the forward blend, interleaved by view: position p = (tile, list offset, list length, first checkpoint) of a tile of view p % V, or a hole
    L.chunk_start = o; o += mgr_align((VT + 1) * 4);
    const size_t rep_u = mgr_rep_units(cap), rep_l = mgr_rep_list(cap), rep_c = mgr_rep_ck(cap);   // (repaired depth-cut tiles append behind the regular entries)
    L.items = o;       o += mgr_align((c / MGR_CHUNK + VT + 1 + rep_c) * 32);   // 32-byte record per (tile, chunk) work item of the backward blend
    L.ckpt = o;        o += mgr_align((c / MGR_CHUNK + 1 + rep_c) * 256 * 16);  // float4 per pixel per checkpoint
    L.keys = o;        o += mgr_align(c * 8);
    L.keys2 = o;       o += mgr_align(c * 8);       // giant tiles: keys regrouped by depth range
    L.groups = o;      o += mgr_align((c / 4096 + 64) * 16);  // (src offset, count) of every depth group
    L.sorted_gid = o;  o += mgr_align((c + rep_l) * 4);
    L.final_T = o;     o += mgr_align(VP * 4);
    L.n_contrib = o;   o += mgr_align(VP * 4);
    L.pair_tag = o;    o += mgr_align(c * 4 + 64);
    L.pair_grad = o;   o += mgr_align(c * 48);
    L.inst_tag = o;    o += mgr_align(VN * 4);      // epoch of the last backward that wrote a record for (view, Gaussian)
    L.inst_grad = o;   o += mgr_align(VN * 128);    // fused backward: 12 floats per (Gaussian, view) + active list
    {
        const size_t VB = (size_t)V * MGR_DB_BUCKETS, bb = (size_t)mgr_bin_block(V, N), nblk = ((size_t)(N > 0 ? N : 1) + bb - 1) / bb;
        L.db_count = o;  o += mgr_align(VB * 4);
        L.db_cursor = o; o += mgr_align(VB * 4);
        L.db_start = o;  o += mgr_align((VB + 1) * 4);
        L.db_nvis = o;   o += mgr_align((size_t)V * 4);
        L.db_bbox = o;   o += mgr_align((size_t)V * 8);      // bounding box of the view's non-empty tiles (x0, y0, w, h)
        L.db_keys = o;   o += mgr_align(VN * 8);      // (depth bits, Gaussian) keys grouped by depth bucket
        L.db_order = o;  o += mgr_align(VN * 4);      // Gaussians of each view in (depth, index) order
        L.db_rec = o;    o += mgr_align(VN * 16);     // their tile rectangles and alive masks, in that order
        L.bin_mat = o;   o += mgr_align((size_t)V * nblk * gx * gy * 4);   // pairs per (block of the order, tile) -> list offsets
    }
    L.tile_bgok = o; o += mgr_align((size_t)V * gx * gy);   // one byte per tile: the caller's image holds the background colour there ("image kept")
    L.tile_rep = o;  o += mgr_align(VT * 4);
    L.rep_unit = o;  o += mgr_align(rep_u * sizeof(MgrRepUnit));
    L.rep_state = o; o += mgr_align(rep_u * 64 * 16);
    L.rep_last = o;  o += mgr_align(rep_u * 64 * 4);
    L.rep_cnt = o;   o += mgr_align(rep_u * 4);
    L.rep_cand = o;  o += mgr_align(rep_u * (size_t)MGR_REP_CAND * 8);
    L.rep_view = o;  o += mgr_align((size_t)V * sizeof(MgrRepView));
    // depth cut: per tile the float bits of the depth behind which the repair of a tile that ran out stops collecting (about
    // MGR_REP_TARGET entries behind the cut; written with the hint by a forward that saw the tile's full list, kept otherwise)
    L.tile_zwin = o; o += mgr_align(VT * 4);
    // instance sort (round 6): the depth range of every per-instance workgroup's visible instances (db_range_reduce, raster_fwd.hip)
    // -> uniform depth buckets over the view's range; per sort item (MGR_DB_ITEM keys of the bucketed order, whole buckets) its first bucket (complement, atomicMax)
    // and the end of its last one -- written by the bucket scan, consumed (left zero) by k_dbin_rank
    L.db_zrange = o; o += mgr_align((size_t)V * (((size_t)(N > 0 ? N : 1) + 511) / 512) * 2 * 4);   // one slot per workgroup of the per-instance kernel (PRE_THREADS = 512)
    L.db_item = o;   o += mgr_align((size_t)V * (((size_t)(N > 0 ? N : 1) + MGR_DB_ITEM - 1) / MGR_DB_ITEM + 1) * 2 * 4);
    L.total = o;
    return L;
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#ifdef __HIPCC__

struct MgrCam {
    float tanfovx, tanfovy;
    float view[16], proj[16];
    float campos[3];
};

// camera record is wave-uniform: these become scalar loads
__device__ __forceinline__ void mgr_load_cam(const float* __restrict__ cams, int v, MgrCam& c) {
    const float* p = cams + (size_t)v * MGR_CAM_FLOATS;
    c.tanfovx = p[0];
    c.tanfovy = p[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        c.view[i] = p[2 + i];
        c.proj[i] = p[18 + i];
    }
    c.campos[0] = p[34];
    c.campos[1] = p[35];
    c.campos[2] = p[36];
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float mgr_dpp(float v) {
    // lanes without a source (or masked rows) receive 0 (old operand)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t mgr_dpp_u(uint32_t v) {   // lanes without a source / masked rows receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// Inclusive wave64 scan of one uint per lane on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then the
// rows' totals across: row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- no LDS round trips, where six
// __shfl_up steps are six ds_bpermute round trips.
__device__ __forceinline__ uint32_t mgr_wave_incl_scan_u32(uint32_t v) {
    v += mgr_dpp_u<0x111>(v);         // row_shr:1
    v += mgr_dpp_u<0x112>(v);         // row_shr:2
    v += mgr_dpp_u<0x114>(v);         // row_shr:4
    v += mgr_dpp_u<0x118>(v);         // row_shr:8
    v += mgr_dpp_u<0x142, 0xa>(v);    // row_bcast:15 -> rows 1, 3
    v += mgr_dpp_u<0x143, 0xc>(v);    // row_bcast:31 -> rows 2, 3
    return v;
}

// Wave64 maximum of one uint per lane, wave-uniform result (DPP inside the rows of 16, the four row results through readlane).
__device__ __forceinline__ uint32_t mgr_wave_max_u32(uint32_t v) {
    v = max(v, mgr_dpp_u<0xb1>(v));    // quad_perm [1,0,3,2]
    v = max(v, mgr_dpp_u<0x4e>(v));    // quad_perm [2,3,0,1]
    v = max(v, mgr_dpp_u<0x141>(v));   // row_half_mirror
    v = max(v, mgr_dpp_u<0x140>(v));   // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// Full wave64 sum; the total is valid in lane 63 only.
__device__ __forceinline__ float mgr_wave_sum63(float v) {
    v += mgr_dpp<0xb1>(v);          // quad_perm [1,0,3,2]
    v += mgr_dpp<0x4e>(v);          // quad_perm [2,3,0,1]
    v += mgr_dpp<0x114>(v);         // row_shr:4
    v += mgr_dpp<0x118>(v);         // row_shr:8
    v += mgr_dpp<0x142, 0xa>(v);    // row_bcast:15 -> rows 1,3
    v += mgr_dpp<0x143, 0xc>(v);    // row_bcast:31 -> rows 2,3
    return v;
}

// Nine independent wave64 sums at once, totals valid in lane 63.  Written with fused
// v_add_f32_dpp (in place: lanes/rows without a DPP source keep their value, i.e. add 0).
// The nine chains are interleaved so that every DPP read of a VGPR is at least two
// instructions after the VALU write of it (the hazard hipcc cannot see inside asm).
#define MGR_DPP9(ctrl)                                                                              \
    asm volatile("s_nop 1\n"                                                                        \
                 "v_add_f32_dpp %0, %0, %0 " ctrl "\n v_add_f32_dpp %1, %1, %1 " ctrl "\n"          \
                 "v_add_f32_dpp %2, %2, %2 " ctrl "\n v_add_f32_dpp %3, %3, %3 " ctrl "\n"          \
                 "v_add_f32_dpp %4, %4, %4 " ctrl "\n v_add_f32_dpp %5, %5, %5 " ctrl "\n"          \
                 "v_add_f32_dpp %6, %6, %6 " ctrl "\n v_add_f32_dpp %7, %7, %7 " ctrl "\n"          \
                 "v_add_f32_dpp %8, %8, %8 " ctrl "\n"                                              \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8))

__device__ __forceinline__ void mgr_wave_sum63_x9(float& a0, float& a1, float& a2, float& a3, float& a4,
                                                  float& a5, float& a6, float& a7, float& a8) {
    MGR_DPP9("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    MGR_DPP9("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    MGR_DPP9("row_shr:4 row_mask:0xf bank_mask:0xf");
    MGR_DPP9("row_shr:8 row_mask:0xf bank_mask:0xf");
    MGR_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf");
    MGR_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm volatile("s_nop 1");
}

// Eight wave64 sums for the price of ~2.3 instructions per value instead of 6: at every level
// two values are folded at once by exchanging lane halves (v_permlane32_swap / v_permlane16_swap,
// new on gfx950) so that each half keeps reducing a different value.  On return the 8 lanes of
// group g = lane >> 3 all hold the wave total of x[MGR_R8_SLOT(g)], MGR_R8_SLOT = {0,4,2,6,1,5,3,7}.
__device__ __forceinline__ void mgr_swap32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void mgr_swap16(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
#define MGR_R8_SLOT(g) ((((g) & 1) << 2) | (((g) & 2)) | (((g) & 4) >> 2))

__device__ __forceinline__ float mgr_wave_reduce8(float x0, float x1, float x2, float x3, float x4, float x5,
                                                  float x6, float x7, int lane) {
    mgr_swap32(x0, x1); const float y0 = x0 + x1;   // lanes 0-31: x0, lanes 32-63: x1
    mgr_swap32(x2, x3); const float y1 = x2 + x3;
    mgr_swap32(x4, x5); const float y2 = x4 + x5;
    mgr_swap32(x6, x7); const float y3 = x6 + x7;
    float a = y0, b = y1, c = y2, d = y3;
    mgr_swap16(a, b); const float z0 = a + b;        // rows: x0, x2, x1, x3
    mgr_swap16(c, d); const float z1 = c + d;        // rows: x4, x6, x5, x7
    const bool hi = (lane & 8) != 0;
    const float send = hi ? z0 : z1, keep = hi ? z1 : z0;
    float w = keep + mgr_dpp<0x128>(send);           // row_ror:8 -> 8-lane groups: z0's row | z1's row
    w += mgr_dpp<0xb1>(w);                           // quad_perm [1,0,3,2]
    w += mgr_dpp<0x4e>(w);                           // quad_perm [2,3,0,1]
    w += mgr_dpp<0x141>(w);                          // row_half_mirror
    return w;
}

__device__ __forceinline__ float mgr_readlane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------
// Interleaved work queue for persistent WAVES.  n_units work units are handed out through MGR_NCTR counters;
// ticket t of counter c is unit t * MGR_NCTR + c.  A wave starts at its home counter and moves on to the next one
// when it runs dry, peeking before it pays for an atomic.  `issue` only sends the atomic; `resolve` reads the
// answer -- in between lies the work that hides its round trip.
// ---------------------------------------------------------------------------
struct MgrQueue {
    uint32_t* ctr;
    uint32_t n_units;
    int c, left;
    __device__ __forceinline__ void init(uint32_t* counters, uint32_t n, int home) {
        ctr = counters; n_units = n; c = home % MGR_NCTR; left = MGR_NCTR;
    }
    __device__ __forceinline__ uint32_t issue(int lane) const {   // raw ticket, valid in lane 0
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(ctr + c * 64, 1u);
        return t;
    }
    __device__ __forceinline__ uint32_t resolve(uint32_t raw, int lane) {   // unit index, or 0xFFFFFFFF when all is handed out
        uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw) * MGR_NCTR + (uint32_t)c;
        while (u >= n_units) {
            if (--left <= 0) return 0xFFFFFFFFu;
            c = (c + 1) % MGR_NCTR;
            const uint32_t cur = __hip_atomic_load(ctr + c * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur * MGR_NCTR + (uint32_t)c >= n_units) { u = 0xFFFFFFFFu; continue; }
            u = (uint32_t)__builtin_amdgcn_readfirstlane((int)issue(lane)) * MGR_NCTR + (uint32_t)c;
        }
        return u;
    }
};

__device__ __forceinline__ int mgr_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// EWA projection rows M0, M1 of (J * Rwv) with the 1.3*tanfov clamp.
__device__ __forceinline__ void mgr_ewa_rows(const MgrCam& c, float W, float H, const float p[3],
                                             float M0[3], float M1[3], float t[3], float& xmul,
                                             float& ymul, float& fx, float& fy) {
#pragma clang fp contract(off)
    const float* v = c.view;
    fx = W / (2.0f * c.tanfovx);
    fy = H / (2.0f * c.tanfovy);
    t[0] = v[0] * p[0] + v[4] * p[1] + v[8] * p[2] + v[12];
    t[1] = v[1] * p[0] + v[5] * p[1] + v[9] * p[2] + v[13];
    t[2] = v[2] * p[0] + v[6] * p[1] + v[10] * p[2] + v[14];
    const float limx = 1.3f * c.tanfovx, limy = 1.3f * c.tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
    const float j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        M0[k] = j00 * v[4 * k + 0] + j02 * v[4 * k + 2];
        M1[k] = j11 * v[4 * k + 1] + j12 * v[4 * k + 2];
    }
}

__device__ __forceinline__ void mgr_sym_mul(const float c6[6], const float m[3], float o[3]) {
#pragma clang fp contract(off)
    o[0] = c6[0] * m[0] + c6[1] * m[1] + c6[2] * m[2];
    o[1] = c6[1] * m[0] + c6[3] * m[1] + c6[4] * m[2];
    o[2] = c6[2] * m[0] + c6[4] * m[1] + c6[5] * m[2];
}

// ---------------------------------------------------------------------------
// exact null-pair culling
//
// A (pixel, Gaussian) evaluation contributes only if alpha = o*exp(-q/2) >= 1/255,
// i.e. q <= qmax = 2 ln(255 o), with q = A dx^2 + 2B dx dy + C dy^2.  If the minimum
// of q over a box of pixel centres exceeds qmax (plus a margin that covers fp32
// rounding of both this bound and the per-pixel test), NO pixel of the box can
// pass the per-pixel test, so dropping the pair changes neither image nor
// gradients.  Non-positive-definite conics or NaNs are never culled.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float mgr_qmax(float opacity) { return 2.0f * __logf(255.0f * opacity); }

// The minimum of the convex quadratic over the box is found coordinate by coordinate, exactly:
//   dyk = the row of the box nearest the centre; on that line q is smallest at dxo = -B dyk / A,
//   and min over the strip as a function of dx is convex with its minimum at dxo, so inside the
//   box it is smallest at dxe = clamp(dxo); there the best dy is dye = clamp(-B dxe / C).
struct MgrCull {
    float cx, cy, A, B, C, nBiA, nBiC, thr;
    bool pd;
};

// (No fma contraction in the three functions below: the test is conservative either way, but its outcome for a tile at the
// threshold must not depend on the loop it was inlined into -- the per-instance kernel has two forms of its cull loop, and a
// null pair listed by one and not by the other shifts the list positions behind it, hence the backward's chunk boundaries
// and its rounding.)
__device__ __forceinline__ MgrCull mgr_cull_init(float cx, float cy, float A, float B, float C, float qmax) {
#pragma clang fp contract(off)
    MgrCull c;
    c.cx = cx; c.cy = cy; c.A = A; c.B = B; c.C = C;
    c.pd = A > 0.0f && C > 0.0f && A * C - B * B > 0.0f;
    c.nBiA = -B * __builtin_amdgcn_rcpf(A);
    c.nBiC = -B * __builtin_amdgcn_rcpf(C);
    c.thr = qmax + 0.01f;
    return c;
}

// row part: dy range of the box and the x of the row-constrained minimum
__device__ __forceinline__ void mgr_cull_row(const MgrCull& c, float y0, float y1, float& dy_lo, float& dy_hi, float& dxo) {
#pragma clang fp contract(off)
    dy_lo = y0 - c.cy;
    dy_hi = y1 - c.cy;
    dxo = c.nBiA * __builtin_amdgcn_fmed3f(0.0f, dy_lo, dy_hi);
}

__device__ __forceinline__ bool mgr_cull_dead(const MgrCull& c, float dy_lo, float dy_hi, float dxo, float x0, float x1) {
#pragma clang fp contract(off)
    const float dxe = __builtin_amdgcn_fmed3f(dxo, x0 - c.cx, x1 - c.cx);
    const float dye = __builtin_amdgcn_fmed3f(c.nBiC * dxe, dy_lo, dy_hi);
    const float t0 = c.A * dxe * dxe, t1 = 2.0f * c.B * dxe * dye, t2 = c.C * dye * dye;
    return c.pd && (t0 + t1 + t2 > c.thr + 1.0e-5f * (t0 + fabsf(t1) + t2));
}

__device__ __forceinline__ bool mgr_box_dead(float cx, float cy, float A, float B, float C, float qmax,
                                             float x0, float y0, float x1, float y1) {
    const MgrCull c = mgr_cull_init(cx, cy, A, B, C, qmax);
    float dy_lo, dy_hi, dxo;
    mgr_cull_row(c, y0, y1, dy_lo, dy_hi, dxo);
    return mgr_cull_dead(c, dy_lo, dy_hi, dxo, x0, x1);
}

// Bounding box (in 8x8-quadrant-local pixel units) of the lanes set in `am`, lane = y*8 + x.
// Returns false when no lane is set.  All scalar (wave-uniform) arithmetic.
__device__ __forceinline__ bool mgr_quad_bbox(unsigned long long am, int& x0, int& y0, int& x1, int& y1) {
    if (am == 0ull) return false;
    y0 = __builtin_ctzll(am) >> 3;
    y1 = (63 - __builtin_clzll(am)) >> 3;
    unsigned int c = (unsigned int)(am | (am >> 32));
    c |= c >> 16;
    c |= c >> 8;
    c &= 0xffu;
    x0 = __builtin_ctz(c);
    x1 = 31 - __builtin_clz(c);
    return true;
}

// sums over aligned groups of G = 2, 4, 8 lanes (DPP quad permutes + row_half_mirror)
template <int G>
__device__ __forceinline__ float grp_sum(float x) {  // all lanes of the group receive the total
    if (G >= 2) x += mgr_dpp<0xb1>(x);   // quad_perm [1,0,3,2]
    if (G >= 4) x += mgr_dpp<0x4e>(x);   // quad_perm [2,3,0,1]
    if (G >= 8) x += mgr_dpp<0x141>(x);  // row_half_mirror
    return x;
}

// 8 values per lane, 8 lanes per group -> lane k of the group returns the group total of x[k]
__device__ __forceinline__ float grp8_reduce_scatter(const float x[8], int vl) {
    const bool b2 = (vl & 4) != 0, b1 = (vl & 2) != 0, b0 = (vl & 1) != 0;
    float y[4], z[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = (b2 ? x[k + 4] : x[k]) + mgr_dpp<0x141>(b2 ? x[k] : x[k + 4]);  // partner 7 - j
#pragma unroll
    for (int k = 0; k < 2; ++k) z[k] = (b1 ? y[k + 2] : y[k]) + mgr_dpp<0x4e>(b1 ? y[k] : y[k + 2]);   // partner j ^ 2
    return (b0 ? z[1] : z[0]) + mgr_dpp<0xb1>(b0 ? z[0] : z[1]);                                        // partner j ^ 1
}

// ---------------------------------------------------------------------------
// Pair-packed staging for the blend kernels.
//
// The blend loops evaluate two list entries per step with packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two lanes of fp32 per VGPR pair).  The survivors of one wave's
// 64-entry batch are therefore written to LDS two by two, field-interleaved, so that one
// ds_read_b128 returns (x_a, x_b, y_a, y_b) etc. already in register-pair order:
//   [0] x_a x_b y_a y_b   [4] A_a A_b B_a B_b   [8] C_a C_b o_a o_b   [12] r_a r_b g_a g_b (forward: r_a g_a r_b g_b)
//   [16] b_a b_b pos_a pos_b
// A, B, C are the conic pre-scaled so that  log2(G) = A dx^2 + B dx dy + C dy^2
// (A = -0.5 log2e conic.x, B = -log2e conic.y, C = -0.5 log2e conic.z): the exponent feeds v_exp_f32
// directly.  Forward and backward evaluate alpha through the same mgr_pair_alpha, so both make
// identical keep/skip decisions.
// ---------------------------------------------------------------------------
typedef float mgr_v2f __attribute__((ext_vector_type(2)));
#define MGR_PAIR_FLOATS 20
#define MGR_LOG2E 1.44269504088896341f
#define MGR_LN2 0.69314718055994531f

// RG_PER_ENTRY (forward blend): [12] r_a g_a r_b g_b -- (r, g) of one entry is a register pair, the colour sum is a
// packed fma with the entry's weight; otherwise (backward) [12] r_a r_b g_a g_b like every other field.
template <bool RG_PER_ENTRY = false>
__device__ __forceinline__ void mgr_pair_store(float* pb, int half, float x, float y, float ca, float cb, float cc,
                                               float op, float r, float g, float b, uint32_t pos) {
    pb[0 + half] = x;
    pb[2 + half] = y;
    pb[4 + half] = (-0.5f * MGR_LOG2E) * ca;
    pb[6 + half] = (-MGR_LOG2E) * cb;
    pb[8 + half] = (-0.5f * MGR_LOG2E) * cc;
    pb[10 + half] = op;
    pb[RG_PER_ENTRY ? 12 + 2 * half : 12 + half] = r;
    pb[RG_PER_ENTRY ? 13 + 2 * half : 14 + half] = g;
    pb[16 + half] = b;
    pb[18 + half] = __uint_as_float(pos);
}

// the second slot of an odd pair: opacity 0 -> alpha 0 -> never valid
template <bool RG_PER_ENTRY = false>
__device__ __forceinline__ void mgr_pair_pad(float* pb) {
#pragma unroll
    for (int f = 0; f < 10; ++f) {
        if (RG_PER_ENTRY && (f == 6 || f == 7)) continue;
        pb[2 * f + 1] = 0.0f;
    }
    if (RG_PER_ENTRY) { pb[14] = 0.0f; pb[15] = 0.0f; }
}

// alpha of two entries at one pixel; pw = log2 of the Gaussian falloff (valid entries have pw <= 0).
// The keep / skip decisions (pw <= 0, alpha >= 1/255) must come out the same in every kernel that inlines this -- the
// forward blend, the backward blend, the debug entry a parity test asks for the device's decision -- so the exponent is
// written with explicit fused multiply-adds and contraction is off: left to the compiler, dx*t + u*dy may become
// fma(dx, t, u*dy) in one kernel and fma(u, dy, dx*t) in another (observed: the debug kernel and the blend disagreed
// on a pair whose alpha equals 1/255 to eight digits).
__device__ __forceinline__ void mgr_pair_alpha_core(const float4 R0, const float4 R1, const float4 R2, mgr_v2f fpx2,
                                                    mgr_v2f fpy2, mgr_v2f& dx, mgr_v2f& dy, mgr_v2f& G, mgr_v2f& al, mgr_v2f& pw) {
#pragma clang fp contract(off)
    const mgr_v2f x2 = {R0.x, R0.y}, y2 = {R0.z, R0.w}, A2 = {R1.x, R1.y}, B2 = {R1.z, R1.w}, C2 = {R2.x, R2.y},
                  o2 = {R2.z, R2.w};
    dx = x2 - fpx2;
    dy = y2 - fpy2;
    mgr_v2f t = A2 * dx;
    t = __builtin_elementwise_fma(B2, dy, t);
    const mgr_v2f u = C2 * dy;
    pw = __builtin_elementwise_fma(dx, t, u * dy);
    // min(2^pw, 1) = 2^min(pw, 0) bit for bit; written as a clamp to [0, 1] it is the `clamp` bit of v_exp_f32 (no v_min)
    G.x = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(pw.x), 0.0f), 1.0f);
    G.y = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(pw.y), 0.0f), 1.0f);
    al = o2 * G;
    al.x = fminf(0.99f, al.x);
    al.y = fminf(0.99f, al.y);
}
// the keep / skip rule of one entry, given its exponent and clamped alpha (one expression for every kernel)
#define MGR_ALPHA_KEPT(pw_, al_) ((pw_) <= 0.0f && (al_) >= 1.0f / 255.0f)

__device__ __forceinline__ void mgr_pair_alpha(const float4 R0, const float4 R1, const float4 R2, mgr_v2f fpx2,
                                               mgr_v2f fpy2, mgr_v2f& dx, mgr_v2f& dy, mgr_v2f& G, mgr_v2f& al,
                                               bool& va, bool& vb, mgr_v2f* pw_out = nullptr) {
    mgr_v2f pw;
    mgr_pair_alpha_core(R0, R1, R2, fpx2, fpy2, dx, dy, G, al, pw);
    va = MGR_ALPHA_KEPT(pw.x, al.x);
    vb = MGR_ALPHA_KEPT(pw.y, al.y);
    if (pw_out) *pw_out = pw;
}

// the same with the decisions as lane masks in scalar registers
__device__ __forceinline__ void mgr_pair_alpha_masks(const float4 R0, const float4 R1, const float4 R2, mgr_v2f fpx2,
                                                     mgr_v2f fpy2, mgr_v2f& al, unsigned long long& ma, unsigned long long& mb) {
    mgr_v2f dx, dy, G, pw;
    mgr_pair_alpha_core(R0, R1, R2, fpx2, fpy2, dx, dy, G, al, pw);
    ma = __builtin_amdgcn_ballot_w64(pw.x <= 0.0f) & __builtin_amdgcn_ballot_w64(al.x >= 1.0f / 255.0f);
    mb = __builtin_amdgcn_ballot_w64(pw.y <= 0.0f) & __builtin_amdgcn_ballot_w64(al.y >= 1.0f / 255.0f);
}

// natural exponential through v_exp_f32 (arguments here lie in [-12, 0])
__device__ __forceinline__ float mgr_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

#endif  // __HIPCC__
