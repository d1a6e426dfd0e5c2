/* manus_hip.h — C ABI of libmanus_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the one hot path of brown-ivl/manus that this project
 * accelerates.  Every entry point is `extern "C"`, takes plain device pointers,
 * sizes and a hipStream_t (passed as void*), and returns 0 or a negative MGR_E*.
 * No torch types appear here; the Python shims in manus_amd/ bind these with
 * ctypes, and INTEGRATION.md shows the binding a MANUS maintainer would add.
 *
 * Reference interfaces replaced (paths relative to the brown-ivl/manus tree):
 *   mgr_raster_forward / mgr_raster_backward
 *       diff_gaussian_rasterization._C.rasterize_gaussians{,_backward}, called by
 *       GaussianRasterizer at src/utils/gaussian_utils.py:393-416 (installed by
 *       setup_env.sh:6; colors_precomp + cov3D_precomp variant only — the one
 *       MANUS uses).
 *   mgr_knn3_mean_dist2
 *       simple_knn._C.distCUDA2, src/models/gaussian.py:4,110.
 *   mgr_skin_weights_fwd/bwd
 *       skinning_weights_from_voxel_grid, src/utils/gaussian_utils.py:167-196
 *       (via HandGaussianModel.get_skin_weights, src/models/hand_gaussian.py:65-76).
 *   mgr_lbs_cov_fwd/bwd
 *       TrainingModule.forward LBS block, src/modules/hand_dynamic.py:106-127, with
 *       GaussianModel.get_covariance, src/models/gaussian.py:49-53,84-93 and
 *       build_rotation/build_scaling_rotation, src/utils/gaussian_utils.py:279-314.
 *   mgr_sh_color_fwd/bwd
 *       calculate_colors_from_sh, src/utils/gaussian_utils.py:431-449 with
 *       eval_sh, src/utils/sh_utils.py:57-104.
 *   mgr_project_points
 *       project_points, src/utils/transforms.py:304-311.
 *   mgr_dilate_mask, mgr_points_outside_mask
 *       dilate_mask / get_points_outside_mask, src/utils/gaussian_utils.py:35-47,101-147.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the name ends in _host.
 *   - All tensors are dense fp32 / int32, caller-owned; outputs are fully
 *     written (no zero-initialisation needed) unless stated.
 *   - "V" = number of camera views batched in one call, "N" = Gaussians.
 *     Per-Gaussian inputs take a view stride in ELEMENTS (floats); 0 means the
 *     same array is shared by every view.
 *   - Cameras: a device array of V records of MGR_CAM_FLOATS floats:
 *       [0] tanfovx [1] tanfovy [2..17] viewmatrix [18..33] projmatrix
 *       [34..36] campos [37..39] unused
 *     with the matrices exactly as MANUS stores them
 *     (`world_view_transform`, `full_proj_transform`, src/utils/cam_utils.py:58-63:
 *     row-major storage of the transposed matrix == column-major math matrix,
 *     element (row r, col c) at [4*c + r]).
 *   - Launches are asynchronous on `stream`; the only host synchronisation is in
 *     the *_sync helpers.  No library-owned device memory.  What the library does
 *     own, all of it host-side:
 *       * a process-wide call counter (the epoch that tags gradient records) and
 *         the profiling switch of mgr_profile_*;
 *       * per device: the "function attributes set" flag (dynamic LDS above 64 KB
 *         is requested on the first forward that runs on a device);
 *       * per (host thread, device), created on first use and only by the per-tile
 *         sort route of the forward (MGR_BINNING=sorted, or a tile grid too large
 *         for the depth-ordered route): ONE non-blocking side stream with a fork
 *         and a join event.  It forks from `stream` and joins it again inside the
 *         call, so the caller sees plain stream order.  The default route, and every
 *         other entry point, launches on `stream` only.
 *     Calls on different devices from different host threads do not share any of
 *     the per-device state (one process per GPU, or one thread per GPU, both work).
 */
#ifndef MANUS_HIP_H
#define MANUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGR_VERSION 100
#define MGR_CAM_FLOATS 40
#define MGR_TILE 16
#define MGR_MAX_BONES 32

enum {
    MGR_OK = 0,
    MGR_EINVAL = -1,   /* bad argument */
    MGR_ENOMEM = -2,   /* workspace too small (see mgr_raster_workspace_bytes) */
    MGR_EHIP = -3,     /* a HIP call failed; text in mgr_last_error() */
    MGR_EOVERFLOW = -4, /* pair capacity exceeded (reported by mgr_raster_status_sync) */
    MGR_ECUT = -6,      /* a forward run with the depth cut (debug bit 8) met a scene its hints no longer fit: its image is
                           incomplete; run it again without the bit (reported by mgr_raster_status_sync) */
    MGR_ETIER = -7      /* a forward told to skip binning launches (debug bits 16 / 32) had a view that needed one: its image
                           is incomplete; run it again without the bits */
};

int mgr_version(void);
/* 0 for the product build.  Non-zero when the library was compiled with one of the instrumentation macros of tools/instr
 * (never by manus_amd.build's defaults): bit 0 = a knock-out that CHANGES RESULTS (BWD_KO, FWD_KO_DEEP: cost bounds
 * only), bit 1 = counters / clocks inside the kernels (MGR_STATS, MGR_TIMELINE, *_PROF: results unchanged, timings
 * perturbed), bit 2 = an alternative code path selected at compile time (FWD_PF1, FWD_LDS_PIPE1, BWD_WLAST_REDUCE, a
 * non-default MGR_BIN_BLOCK: results unchanged).  bench.py labels such a run and refuses to call it the headline; the parity
 * block does not run on a library that reports bit 0. */
int mgr_build_variant(void);
/* Thread-local text of the last error returned on this host thread. */
const char* mgr_last_error(void);

/* ------------------------------------------------------------------------
 * Rasterizer (tile binning, per-tile depth sort, alpha compositing, backward)
 * ------------------------------------------------------------------------ */

/* Bytes of workspace needed for V views of N Gaussians at W x H with room for
 * `pair_capacity` (Gaussian, tile) pairs summed over all views.  The workspace
 * must be zero-filled once when (re)allocated, and again before it is reused
 * with a different (V, N, W, H, pair_capacity): it carries counters, tags and
 * per-tile state from one call to the next (a forward leaves its tile counters
 * zeroed for the following one instead of clearing them per call); otherwise it
 * is opaque state that links a forward call to its backward call. */
size_t mgr_raster_workspace_bytes(int V, int N, int W, int H, int64_t pair_capacity);

/* Forward.  out_color: (V,3,H,W).  radii: (V,N) int32.  bg: 3 floats.
 * means3D (N,3), cov3D (N,6) packed [xx,xy,xz,yy,yz,zz], colors (N,3),
 * opacity (N) — each with its per-view stride.
 * If the number of pairs exceeds pair_capacity the image is still written but
 * is incomplete and the overflow flag is raised: check with
 * mgr_raster_status_sync and retry with a larger workspace.
 * debug (here and in mgr_views_forward) is a bit set: 1 = synchronise and check after every kernel (upstream's
 * `debug=True`); 2 = stop before the blend (projection and binning only: radii, the pair total and the tile lists are
 * final, out_color is not written); 4 = the blend only, after a call with bit 2 on the same workspace and arguments.
 * Bits 2 / 4 let a caller put work that needs the tile lists but not the image next to the blend (the image loss's
 * span list, mgr_image_loss_tiles_list).  The tile lists come from the depth-ordered binning; MGR_BINNING=sorted in
 * the environment selects the per-tile sorts instead (identical lists).
 * 8 (mgr_views_forward only) = depth cut.  Every forward leaves, per tile whose pixels all saturated, the depth in front
 * of which they had all stopped plus a margin; with bit 8 the next forward on the same workspace leaves the instances
 * behind that depth out of the tile's list (they lie behind every pixel's stop: image, n_contrib and gradients are bit
 * for bit those of the full lists, but the binning handles a fraction of the pairs).  Only valid when that previous
 * forward rendered the SAME views (camera + pose) of a model that has moved little since; if a cut list runs out under
 * a pixel that has not saturated the forward raises the overflow word's bit 1 (mgr_raster_status_sync: MGR_ECUT) and
 * the caller runs it again without bit 8.  Pass the bit to both calls of a forward split with bits 2 / 4.  No
 * counterpart upstream (the reference renders one view per step and re-bins everything).
 * 2048 (with bit 8) = repair on the device.  A tile whose cut list runs out under an unsaturated pixel is completed by
 * two kernels behind the blend instead of flagging the forward: the instances the cut dropped from that tile are found
 * (one pass over the view's rectangles), sorted by (depth, index) -- the tail of the tile's full list --, appended
 * behind the regular lists, and the walks of the tile's unsaturated quadrants continue from their saved state; the
 * backward's work items of such a tile point at the appended entries / checkpoints.  Image, n_contrib and gradients
 * stay bit for bit those of the full lists and NO re-run is needed; MGR_ECUT is only raised when a capacity of the repair
 * is exceeded (repaired quadrants, 256 tiles per view, 8192 entries behind the cut of one tile, the appended entries: all
 * sized from pair_capacity), or when a hinted tile ends up with no list at all.  A tile that ran out gets no hint for the
 * next mgr_raster_set_cut_penalty forwards (the same few tiles at the rim of the saturating region otherwise run out
 * step after step under a moving model).  The status mirror's overflow word carries the number of repaired quadrants
 * of the forward in its bits 16..31.
 * 16 / 32 = skip the binning launches that only serve views whose box of non-empty tiles has more than 2048 / has
 * 1537..2048 tiles (they hold more LDS per workgroup; three launches of ~6 us each that do nothing for smaller boxes).
 * mgr_raster_status_tiers_sync reports which of them a forward needed (bit 0 / bit 1); pass the bits for the tiers the
 * previous forward did not need.  A view that needs a skipped launch raises the overflow word's bit 2 (MGR_ETIER).
 * 128 = skip the launch behind the instance sort.  Since round 6 the (depth, index) keys of a view are sorted in items of
 * ~768 keys, one workgroup each (k_dbin_rank: depth buckets uniform over the depth range of the view's visible instances in
 * this forward); an item of more than 2048 keys -- a dense depth slice -- is left to a radix launch behind, which returns at
 * once when there is none.  Bit 128 omits that launch (mgr_raster_status_tiers_sync reported no item near the limit for the
 * previous forward: bits 8..15 of its tiers word for the usual instantiation, bits 16..23 for the one bit 256 asks for);
 * an item that needs it then raises the overflow word's bit 2 (MGR_ETIER:
 * run the forward again without the bit), like a skipped tile-box tier.  256 = k_dbin_rank's instantiation for items of up
 * to 3072 keys (the previous forward met items of more than 2048: bits 24..30 of the tiers word) -- ~4 us slower for all its
 * items, but the dense slice no longer waits for the launch behind (33 us).  4096 = k_bin_scatter's lane-spreading
 * instantiation (the previous forward met rectangles of more than 64 tiles: bit 2 of the tiers word): a batch that holds such
 * a rectangle is spread over lanes, one row piece per lane, instead of going instance by instance (correct either way;
 * cameras close to the hand: 1.2 -> 0.87 ms).  Without the bit the producer is the plain one of round 5. */
int mgr_raster_forward(int V, int N, int W, int H, const float* cams, const float* bg,
                       const float* means3D, int64_t stride_means3D, const float* cov3D,
                       int64_t stride_cov3D, const float* colors, int64_t stride_colors,
                       const float* opacity, int64_t stride_opacity, float* out_color,
                       int32_t* radii, void* workspace, size_t workspace_bytes,
                       int64_t pair_capacity, int debug, void* stream);

/* Backward of the forward that last used `workspace` (same inputs again, plus the
 * image that forward produced: out_color (V,3,H,W)).
 * dL_dcolor: (V,3,H,W).  Outputs, all fully written:
 * dL_dmeans3D (V,N,3), dL_dmeans2D (V,N,3) (z = 0; x,y in NDC-scaled pixel units
 * 0.5*W, 0.5*H as the reference's densification statistic expects,
 * src/models/gaussian.py:335-338), dL_dcolors (V,N,3), dL_dopacity (V,N),
 * dL_dcov3D (V,N,6) (off-diagonals carry the factor 2 of the symmetric pack). */
int mgr_raster_backward(int V, int N, int W, int H, const float* cams, const float* bg,
                        const float* means3D, int64_t stride_means3D, const float* cov3D,
                        int64_t stride_cov3D, const float* colors, int64_t stride_colors,
                        const float* opacity, int64_t stride_opacity, const float* out_color,
                        const float* dL_dcolor, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                        float* dL_dopacity, float* dL_dcov3D, void* workspace,
                        size_t workspace_bytes, int64_t pair_capacity, int debug, void* stream);

/* ------------------------------------------------------------------------
 * Fused articulated path (training engine): canonical parameters in, image out.
 * One launch chain per V views = mgr_lbs_cov_fwd + mgr_sh_color_fwd + sigmoid +
 * mgr_raster_forward, without writing posed means / covariances / transforms /
 * colours to HBM; the backward returns the leaf gradients of the six MANUS
 * parameter tensors (src/models/gaussian.py:34-39) summed over the views, the
 * skin-weight gradient (feed it to mgr_skin_weights_bwd), and the densification
 * statistics of src/models/gaussian.py:335-338 / src/utils/gaussian_utils.py:469-471.
 *   xyz (N,3), log_scale (N,3) = _scaling, rot (N,4) = _rotation (raw),
 *   opacity_logit (N) = _opacity, f_dc (N,1,3) = _features_dc, f_rest (N,15,3) = _features_rest,
 *   skin_w (n_articulated,B) or NULL (static object), transforms (V,B,16): one pose per view.
 *   n_articulated <= N: the first n_articulated Gaussians are skinned, the rest are static (identity
 *   transform, no skin-weight row): the hand+object concatenation of src/modules/composite.py:50-59.
 *   d_skin_w is (n_articulated,B).
 *   sh_half != 0: f_rest points to the fp16 storage copy of _features_rest made by mgr_sh_to_half -- (N,48) halves,
 *   16-byte aligned rows, 45 used (BASELINE config 5 "fp16 SH coeffs"; the reference has no counterpart, everything
 *   there is fp32).  Arithmetic and d_f_rest stay fp32 (N,15,3); the optimizer owns the fp32 master copy.
 * stat_grad2d (N): sum over views of ||dL/dmeans2D[:, :2]|| * grad2d_scale,
 * stat_vis (N): number of views with radius > 0, stat_radii (N): max radius (any may be NULL).
 * ------------------------------------------------------------------------ */
int mgr_views_forward(int V, int N, int B, int n_articulated, int sh_half, int W, int H, const float* cams, const float* bg,
                      const float* xyz, const float* log_scale, const float* rot,
                      const float* opacity_logit, const float* f_dc, const float* f_rest,
                      const float* skin_w, const float* transforms, float* out_color, int32_t* radii,
                      void* workspace, size_t workspace_bytes, int64_t pair_capacity, int debug,
                      void* stream);
int mgr_views_backward(int V, int N, int B, int n_articulated, int sh_half, int W, int H, const float* cams, const float* bg,
                       const float* xyz, const float* log_scale, const float* rot,
                       const float* opacity_logit, const float* f_dc, const float* f_rest,
                       const float* skin_w, const float* transforms, const int32_t* radii,
                       const float* out_color, const float* dL_dcolor, float grad2d_scale, float* d_xyz,
                       float* d_log_scale, float* d_rot, float* d_opacity_logit, float* d_f_dc,
                       float* d_f_rest, float* d_skin_w, float* stat_grad2d, float* stat_vis,
                       int32_t* stat_radii, void* workspace, size_t workspace_bytes,
                       int64_t pair_capacity, int debug, void* stream);

/* debug bit 1024 of mgr_views_forward / mgr_raster_forward = "image kept": the caller vouches that out_color is the image the
 * previous complete forward on this workspace wrote, untouched since.  A tile that held the background colour then and is
 * empty again is not written again (85 % of the tiles of a capture-like frame are empty: 170 MB of stores at eight 1080p
 * views); the image is identical.  Honoured only when the header says so: that forward was the last one binned on this
 * workspace, wrote this very buffer, with this background colour -- otherwise every empty tile is written as without the bit.
 * (Without the bit the background of the empty tiles is written by extra workgroups of the instance sort's launch when the
 * depth-ordered binning runs, by the blend otherwise; MANUS_BG_FILL=blend in the environment: always by the blend.) */

/* debug of mgr_views_backward: bit 1 as above.  Bit 512 = "outputs kept": the caller vouches that the leaf-gradient buffers
 * (d_xyz ... d_skin_w, stat_grad2d) are the ones the previous mgr_views_backward on this workspace wrote, untouched since
 * (persistent .grad-like tensors).  The backward then zeroes only the rows that call wrote and this one does not, instead
 * of every row of every buffer (97 MB of stores on the bench step); the results are identical.  Honoured only when the
 * workspace's row state is that previous call's and describes these very buffers (the library checks a call counter and
 * d_xyz; V <= 8, 3..8 views, run lists on) -- otherwise every row is zeroed as without the bit. */

/* Device pointers (into the workspace) to the compacted list of Gaussians that received a gradient in the last
 * mgr_views_backward and to its length; V <= 8. */
int mgr_views_active_list(void* workspace, int V, int N, int W, int H, int64_t pair_capacity, const uint32_t** list,
                          const uint32_t** count);

/* Lane layout of the fused per-instance backward at 3..8 views per group (lane groups of four or eight; process-wide; returns
 * the previous setting).
 * on (default; MANUS_INST_RUNS=0 in the environment starts with it off): an active Gaussian takes 8 / 4 / 2 lanes by the
 * number of its views that hold pair-gradient records (41 % of the (active Gaussian, view) lanes did on the bench step), and
 * the gather walks a compacted list of the instances with records;
 * off: always one lane per view.  The per-view contributions are summed in ascending view order either way, over a tree of
 * the views with records (on) or of all views (off): the results agree to rounding (tests/test_gpu_fused.py), each is
 * bit-reproducible.  No reference counterpart (there, autograd accumulates the views' contributions into .grad). */
int mgr_views_backward_run_lists(int on);

/* f_rest (N,45) fp32 -> out_half (N,48) fp16 (round to nearest even, 3 halves of zero padding per row). */
int mgr_sh_to_half(int N, const float* f_rest, void* out_half, void* stream);

/* Debug/test: byte offsets of the workspace regions, in the order header, grec, depth, rect,
 * alive, pair_off, tile_count, tile_start, tile_cursor, tile_done, tile_queue, chunk_start, items,
 * ckpt, keys, sorted_gid, final_T (reserved, not written), n_contrib, pair_tag, pair_grad, total, inst_grad
 * (fused backward: per (Gaussian, view-lane) the 9 gathered blend sums, 12 floats each, then the active list),
 * inst_tag, db_nvis (per view: instances with at least one non-null tile), db_bbox (per view: ushort4 x0, y0, w, h of the
 * non-empty tiles), db_order (per view: those instances in (depth, index) order) -- these three belong to the
 * depth-ordered binning --, tile_zcut, tile_zused, tile_qend (depth cut: hint of the next forward / applied / end of the
 * tile's walks), tile_rep, rep_unit, rep_cnt (repair of depth-cut tiles: owner unit + 1 per tile, the 64-byte unit records,
 * entries found behind the cut per owner unit).  Returns the count. */
int mgr_raster_layout(int V, int N, int W, int H, int64_t pair_capacity, size_t* out, int n_out);
/* Debug/test: stride in bytes of the per-(view, Gaussian) records of the workspace's `grec` region (x, y, conic A B C, opacity,
 * r g b, pair-slot base, rectangle width: twelve 4-byte words, then padding to the stride). */
int mgr_raster_record_bytes(void);

/* Blocking read-back of the workspace header after a forward: total number of
 * (Gaussian, tile) pairs (`num_rendered`, summed over views) and the overflow
 * flag.  Returns MGR_EOVERFLOW when the flag is set. */
/* Margins of the depth-cut hints the forwards of this host thread leave (debug bit 8): a tile keeps the entries its walks
 * used plus max(min_entries, frac_entries x that many), and at least the depth of the last one plus depth_range_frac of the
 * walked depth range plus depth_rel of that depth.  interior_only: hints only for tiles whose eight neighbours saturated
 * too (a silhouette tile stops saturating when an edge moves by a fraction of a pixel).  Defaults 0.125, 64, 0.0625, 2e-4, 0.
 * Wider margins: more pairs binned, fewer forwards flagged MGR_ECUT when the model moves between two forwards of a view. */
int mgr_raster_set_cut_margin(float frac_entries, int min_entries, float depth_range_frac, float depth_rel, int interior_only);
/* Forwards (of this host thread) for which a tile whose cut list ran out receives no hint; default 16, 0 = none. */
int mgr_raster_set_cut_penalty(int forwards);
/* Status without a copy: `host_words` points to 4 uint32 of host memory the device can write (hipHostMalloc / pinned
 * memory); the next forward that runs the blend on `workspace` writes (pair total, overflow word, binning tiers, 1) there
 * from its last kernel.  An event recorded behind that forward then tells the host when the words are valid -- no
 * device-to-host copy, no host synchronisation of the stream.  One shot per call; nullptr withdraws a pending mirror.  The
 * caller zeroes word 3 beforehand and keeps the memory alive until it has read it.  (The reference's rasterizer returns
 * num_rendered through a blocking read-back, SURVEY App. A.) */
int mgr_raster_set_status_mirror(const void* workspace, void* host_words);
/* mgr_raster_status_sync plus `tiers` (see debug bits 16 / 32 / 128 / 256 / 4096 of the forward): bit 0 = a view's tile box
 * had more than 2048 tiles, bit 1 = one had 1537..2048, bit 2 = rectangles of more than 64 tiles were met (pass bit 4096
 * next time); bits 8..15 / 16..23 = items of the instance sort beyond 13/16 of k_dbin_rank's 2048 / 3072 keys (capped at
 * 255): zero in the field that belongs to the instantiation the next forward asks for lets it pass bit 128; bits 24..30 =
 * items beyond 2048 keys (pass bit 256 next time). */
int mgr_raster_status_tiers_sync(const void* workspace, int64_t* num_pairs, int32_t* overflow, int32_t* tiers,
                                 void* stream);
int mgr_raster_status_sync(const void* workspace, int64_t* num_pairs, int32_t* overflow,
                           void* stream);

/* Debug/test access to the binning state of view v after a forward (blocking):
 * tile_ranges_host (tiles,2) int32 [start,end) into the global sorted list,
 * point_list_host up to `max_pairs` Gaussian indices in blend order. */
int mgr_raster_debug_binning_sync(const void* workspace, int V, int N, int W, int H,
                                  int64_t pair_capacity, int view, int32_t* tile_ranges_host,
                                  int32_t* point_list_host, int64_t max_pairs, void* stream);

/* Debug/test: the blend kernels' own evaluation of alpha for n (Gaussian record, pixel) pairs -- the device function
 * both blend kernels call, exp through v_exp_f32 in the log2 domain.  rec (n,6) = pixel centre x, y, conic A, B, C,
 * opacity; px, py (n) pixel coordinates; out: alpha (n) and valid (n) = 1 when the pair passes the kernels' tests
 * (power <= 0 and alpha >= 1/255).  Lets a parity test decide which side of the threshold the
 * kernels took for pairs that sit within rounding of it.  All pointers are device pointers. */
int mgr_debug_pair_alpha(int n, const float* rec, const int32_t* px, const int32_t* py, float* alpha, int32_t* valid,
                         void* stream);

/* ------------------------------------------------------------------------
 * Articulation: skin weights, LBS of means + covariances, SH colour
 * ------------------------------------------------------------------------ */

/* Trilinear sample (align_corners=True, zero padding) of a channel-last grid
 * (D,H,W,B) at u = (xyz - center)/scale, u=(x,y,z) indexing (W,H,D), then
 * w /= sum(w) without epsilon (0/0 -> NaN exactly like the reference).
 * out_w: (N,B).  B <= MGR_MAX_BONES.  grid_stride = floats per voxel: B (the
 * reference layout) or 24 (channels zero-padded to 24, 16-byte aligned base: the
 * fast path with float4 gathers; needs B <= 24). */
int mgr_skin_weights_fwd(int N, const float* xyz, const float* grid, int D, int H, int W, int B,
                         int grid_stride, const float* center3, const float* scale3, float* out_w,
                         void* stream);
/* dL_dxyz (N,3) written, or added to when accumulate != 0 (the training engine adds the skin-weight path
 * to the gradient mgr_views_backward has already written for the same xyz leaf). */
int mgr_skin_weights_bwd(int N, const float* xyz, const float* grid, int D, int H, int W, int B,
                         int grid_stride, const float* center3, const float* scale3,
                         const float* dL_dw, float* dL_dxyz, int accumulate, void* stream);
/* The same for the Gaussians index[0 .. *index_count) only (device list + device count <= max_count, e.g. the active
 * list of mgr_views_active_list: rows of dL_dw that received nothing are zero and contribute nothing); entries >= N
 * (static Gaussians of a composite) are skipped; always accumulates. */
int mgr_skin_weights_bwd_indexed(int N, const float* xyz, const float* grid, int D, int H, int W, int B,
                                 int grid_stride, const float* center3, const float* scale3, const float* dL_dw,
                                 float* dL_dxyz, const uint32_t* index, const uint32_t* index_count, int max_count,
                                 void* stream);

/* LBS for P poses.  transforms: (P,B,16) row-major 4x4 bone transforms
 * T_b = posed_b * inv(rest_b) (+ identity background).  skin_w (N,B) or NULL for
 * the static-object path (identity transform: posed = xyz, cov = Sigma).
 * Outputs: posed_xyz (P,N,3), posed_cov (P,N,6), tf (P,N,12) = rows 0..2 of the
 * blended 4x4 (may be NULL). */
int mgr_lbs_cov_fwd(int P, int N, int B, const float* xyz, const float* log_scale,
                    const float* rot, const float* skin_w, const float* transforms,
                    float* posed_xyz, float* posed_cov, float* tf, void* stream);
/* Backward.  dL_dtf (P,N,12) may be NULL.  Outputs fully written, summed over
 * poses: dL_dxyz (N,3) (direct path only — the skin-weight path is dL_dw),
 * dL_dlog_scale (N,3), dL_drot (N,4), dL_dw (N,B) (NULL when skin_w is NULL). */
int mgr_lbs_cov_bwd(int P, int N, int B, const float* xyz, const float* log_scale,
                    const float* rot, const float* skin_w, const float* transforms,
                    const float* dL_dposed_xyz, const float* dL_dposed_cov, const float* dL_dtf,
                    float* dL_dxyz, float* dL_dlog_scale, float* dL_drot, float* dL_dw,
                    void* stream);

/* SH degree-3 colour for V views.  sh: (N,16,3).  If tf != NULL (V,N,12 with
 * stride_tf floats per view, 0 = shared) the camera is pulled back to canonical
 * space: dir = xyz - inv(tf)*campos; else dir = xyz - campos, where xyz has
 * stride_xyz floats per view.  colors (V,N,3) = max(sh2rgb + 0.5, 0). */
int mgr_sh_color_fwd(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                     const float* tf, int64_t stride_tf, const float* cams, float* colors,
                     void* stream);
/* Outputs fully written: dL_dsh (N,16,3) summed over views, dL_dxyz (V,N,3) per
 * view, dL_dtf (V,N,12) per view (NULL when tf is NULL). */
int mgr_sh_color_bwd(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                     const float* tf, int64_t stride_tf, const float* cams,
                     const float* dL_dcolors, float* dL_dsh, float* dL_dxyz, float* dL_dtf,
                     void* stream);

/* The same four with the rows of tf / dL_dtf `tf_row_floats` apart: 12 (as above) or 16 = the reference's (N,4,4) layout
 * (TrainingModule.forward returns the blended transforms as 4x4, src/modules/hand_dynamic.py:128-137, and render_gaussians
 * takes them as such, src/utils/gaussian_utils.py:431-449): mgr_lbs_cov_fwd_rows then writes the constant last row
 * (0,0,0,1) too, mgr_sh_color_bwd_rows writes zeros into the last row of dL_dtf, the readers skip it -- no torch.cat /
 * slice copy between the two operators and none in their backward (round 6). */
int mgr_lbs_cov_fwd_rows(int P, int N, int B, const float* xyz, const float* log_scale,
                         const float* rot, const float* skin_w, const float* transforms,
                         float* posed_xyz, float* posed_cov, float* tf, int tf_row_floats, void* stream);
int mgr_lbs_cov_bwd_rows(int P, int N, int B, const float* xyz, const float* log_scale,
                         const float* rot, const float* skin_w, const float* transforms,
                         const float* dL_dposed_xyz, const float* dL_dposed_cov, const float* dL_dtf, int tf_row_floats,
                         float* dL_dxyz, float* dL_dlog_scale, float* dL_drot, float* dL_dw, void* stream);
int mgr_sh_color_fwd_rows(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                          const float* tf, int64_t stride_tf, int tf_row_floats, const float* cams, float* colors,
                          void* stream);
int mgr_sh_color_bwd_rows(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                          const float* tf, int64_t stride_tf, int tf_row_floats, const float* cams,
                          const float* dL_dcolors, float* dL_dsh, float* dL_dxyz, float* dL_dtf,
                          void* stream);

/* Host glue of the reference-shaped route, one launch each (no reference kernel counterpart: the reference does both with
 * a dozen small torch ops per step).
 * mgr_pack_camera: the (MGR_CAM_FLOATS = 40)-float camera row every kernel here reads, from the fields of
 *   GaussianRasterizationSettings (src/utils/gaussian_utils.py:378-391): [tanfovx, tanfovy, viewmatrix 16, projmatrix 16,
 *   campos 3, 0 0 0]; the matrices as the reference stores them (row-major of the transposed = column-major float[16],
 *   SURVEY App. A).  view16 / proj16 / campos3 are DEVICE pointers and may be NULL (zeros); the scalars travel as kernel
 *   arguments, so no host-to-device copy is made.
 * mgr_bone_transforms: T_b = posed_b @ inv(rest_b) for B bones (4x4 row-major each), + one identity row when
 *   `background` (src/modules/hand_dynamic.py:93-102); out (B + background, 4, 4). */
int mgr_pack_camera(float tanfovx, float tanfovy, const float* view16, const float* proj16, const float* campos3,
                    float* out40, void* stream);
int mgr_bone_transforms(int B, int background, const float* posed, const float* rest, float* out, void* stream);

/* uv = (K*E*[x;1])[:2]/z for N points; K (3,3), E (3,4) row-major. */
int mgr_project_points(int N, const float* xyz, const float* K9, const float* E12, float* uv,
                       void* stream);

/* Segmentation-mask pruning test of on_after_backward (src/modules/hand_dynamic.py:193-209,
 * src/modules/object.py:66-72).
 * mgr_dilate_mask: dilate_mask, src/utils/gaussian_utils.py:35-47 (conv2d with a kernel_size^2 box of
 *   ones, zero padding, > 0) on a (H,W) byte mask (non-zero = set); scratch and out are (H,W) bytes.
 * mgr_points_outside_mask: get_points_outside_mask, src/utils/gaussian_utils.py:101-147:
 *   out[i] = !mask[int(clamp(v_i, 0, H-1))][int(clamp(u_i, 0, W-1))] with (u,v) = project_points(xyz_i);
 *   when any of the n_keypoints keypoints (may be 0 / NULL) lands outside the mask, all out[i] = 0
 *   (:125-131).  mask: the (already dilated, when the caller asks for dilate=True) (H,W) byte mask. */
int mgr_dilate_mask(int H, int W, int kernel_size, const uint8_t* mask, uint8_t* scratch, uint8_t* out,
                    void* stream);
int mgr_points_outside_mask(int N, const float* xyz, const float* K9, const float* E12, int H, int W,
                            const uint8_t* mask, int n_keypoints, const float* keypoints, uint8_t* out,
                            void* stream);
/* out[i] = mean_k |xyz_i - keypoint_k| > thresh: the keypoint-distance pruning test,
 * src/modules/hand_dynamic.py:210-218 (torch.cdist(posed_xyz, keypoints).mean(1) > 0.2). */
int mgr_keypoint_far_mask(int N, const float* xyz, int n_keypoints, const float* keypoints, float thresh,
                          uint8_t* out, void* stream);

/* ------------------------------------------------------------------------
 * simple-knn: mean squared distance to the 3 nearest other points
 * ------------------------------------------------------------------------ */
size_t mgr_knn3_workspace_bytes(int N);
int mgr_knn3_mean_dist2(int N, const float* xyz, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Image loss used by the benchmark step: L = mean|a-b| over (V,3,H,W);
 * writes dL/da = sign(a-b) * scale and accumulates sum|a-b| into loss_sum[0].
 * (src/utils/loss_utils.py:22-27 with src/modules/base.py:329-331.)
 * ------------------------------------------------------------------------ */
int mgr_l1_loss_grad(int64_t count, const float* a, const float* b, float scale, float* dL_da,
                     float* loss_sum, void* stream);

/* ------------------------------------------------------------------------
 * Training image loss, forward + backward in one pass:
 *   L = grad-scaled  w_l1 * sum|pred - target|  +  w_ssim * (- sum ssim_map)
 * over images in the rasterizer's (V,3,H,W) layout.  Replaces l1_loss
 * (src/utils/loss_utils.py:22-27) and ssim (src/utils/loss_utils.py:57-97) as
 * called by loss_func (src/modules/base.py:323-365) together with their
 * autograd backward.  The reference evaluates ssim() on HWC tensors, so its
 * 11x11 window runs over the (W,3) plane of every image row (groups = H,
 * loss_utils.py:58); that is the statistic computed here.
 *   dL_dpred (V,3,H,W) = grad_scale * (w_l1 * sign(pred - target) - w_ssim * d(sum ssim_map)/dpred)
 *   sums[0] = sum|pred - target|, sums[1] = sum of the ssim map (both over V*3*H*W values),
 *   sums[2] = grad_scale * (w_l1 * sums[0] - w_ssim * sums[1]) + loss_offset  (the loss value whose
 *             gradient dL_dpred is; loss_offset carries the constant of "1 - ssim"),
 * so mean L1 = sums[0]/(V*3*H*W) and the reference's ssim(...) of one view = sums[1]/(3*H*W).
 * workspace: mgr_image_loss_workspace_bytes(V,H,W) bytes of scratch (per-workgroup sums).
 * ------------------------------------------------------------------------ */
size_t mgr_image_loss_workspace_bytes(int V, int H, int W);
int mgr_image_loss(int V, int H, int W, const float* pred, const float* target, float w_l1, float w_ssim,
                   float grad_scale, float loss_offset, float* dL_dpred, float* sums, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same loss when `pred` is the image mgr_raster_forward / mgr_views_forward has just rendered with background bg3
 * and tile_start points at that forward's tile-list offsets (workspace + mgr_raster_layout()[7]; V * T + 1 words,
 * T = ceil(W/16) * ceil(H/16)).  Where every tile under a span of the loss holds no Gaussian the rendered pixels ARE
 * the background colour, so the span only has to be compared with the target's background: the rendered image is not
 * read there, and no zero gradient is written.  sums are those of mgr_image_loss.  dL_dpred is written wherever a tile
 * holds a Gaussian (everywhere the backward pass reads it) and wherever the target differs from the background; it is
 * left untouched under empty tiles whose target is background. */
int mgr_image_loss_tiles(int V, int H, int W, const float* pred, const float* target, const float* bg3,
                         const uint32_t* tile_start, float w_l1, float w_ssim, float grad_scale, float loss_offset,
                         float* dL_dpred, float* sums, void* workspace, size_t workspace_bytes, void* stream);
/* mgr_image_loss_tiles in two calls.  _list builds the span list: it needs the forward's tile offsets but not its
 * image, so it can run on another stream while the forward blend runs (mgr_views_forward / mgr_raster_forward with
 * debug bit 1 = "stop before the blend", then bit 2 = "the blend only").  _finish does the rest on that list (same
 * workspace; the caller orders it after both the list and the blend). */
int mgr_image_loss_tiles_list(int V, int H, int W, const float* target, const float* bg3, const uint32_t* tile_start,
                              void* workspace, size_t workspace_bytes, void* stream);
/* The span list without reading the targets.  mgr_image_loss_target_map computes, once per set of target images and
 * background colour, the column masks the list is derived from (mgr_image_loss_target_map_words(V,H,W) uint32);
 * mgr_image_loss_tiles_list_mapped then builds the list of mgr_image_loss_tiles_list from those masks and the forward's tile
 * offsets.  Same list, same sums and gradients; a target image is a constant of its view (the reference reads it from the
 * dataset every step, src/modules/base.py:323-365). */
size_t mgr_image_loss_target_map_words(int V, int H, int W);
int mgr_image_loss_target_map(int V, int H, int W, const float* target, const float* bg3, uint32_t* map, void* stream);
int mgr_image_loss_tiles_list_mapped(int V, int H, int W, const uint32_t* map, const float* bg3, const uint32_t* tile_start,
                                     void* workspace, size_t workspace_bytes, int workspace_kept, void* stream);
/* workspace_kept != 0: the workspace was zero-filled when allocated and has only been used by list / finish pairs since (the
 * finish pass leaves its list counter zero): the 4-byte memset per call is skipped. */
/* The same list built by the forward itself: attached to `raster_workspace`, the next forward (mgr_views_forward /
 * mgr_raster_forward) that runs its blend on that workspace builds the list in extra workgroups of its last kernel, from
 * its own tile offsets (one shot, like mgr_raster_set_status_mirror; map == NULL withdraws a pending attachment).
 * `loss_workspace` must be a kept one (zero-filled once, only ever used by list / finish pairs: nothing is cleared here);
 * mgr_image_loss_tiles_finish follows as after mgr_image_loss_tiles_list_mapped.  V, H, W must be the forward's.  (Round 6:
 * as a launch of its own the list was 8 us in the chain of small kernels between the forward blend and the loss.) */
int mgr_views_forward_attach_loss_list(const void* raster_workspace, int V, int H, int W, const uint32_t* map,
                                       void* loss_workspace, size_t loss_workspace_bytes);
int mgr_image_loss_tiles_finish(int V, int H, int W, const float* pred, const float* target, float w_l1, float w_ssim,
                                float grad_scale, float loss_offset, float* dL_dpred, float* sums, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Optimizer step and densification of the Gaussian parameter model
 * (src/models/gaussian.py:128-338).
 *
 * mgr_adam_step: one torch.optim.Adam update (betas, eps as given; no weight
 * decay, no amsgrad; the reference uses Adam(l, lr=0, eps=1e-15),
 * gaussian.py:142) of `n_groups` <= 8 parameter groups in one launch.
 * counts / lrs are host arrays; params, grads, exp_avg, exp_avg_sq are host
 * arrays of device pointers (fp32, counts[k] elements each).  `step` is the
 * 1-based step number of this update (bias correction).
 *
 * mgr_reset_opacity: reset_opacity, gaussian.py:148-165 (opacity <-
 * inverse_sigmoid(min(sigmoid(opacity), 0.01)), both moments zeroed).
 *
 * mgr_densify_plan + mgr_densify_apply: densify_and_prune, gaussian.py:310-333
 * (clone :288-308, split with N=2 :254-286, prune :183-200, optimizer-state
 * surgery :148-252).  plan classifies the N Gaussians from the densification
 * statistics (grad = accum / denom, NaN -> 0), scans, builds the source map in
 * `workspace` and returns (blocking) counts_host[5] = {kept originals, kept
 * clones, split-selected, kept split parents, M = new number of Gaussians}.
 * apply writes the M new rows of the six leaves in group order (xyz 3, f_dc 3,
 * f_rest 45, opacity 1, scaling 3, rotation 4), of both Adam moments (zero for
 * new rows) and of the skin weights (N,B) (may be NULL), in the reference's
 * order [kept originals | clones | split copies 0 | split copies 1].
 * noise: standard normals (2 * n_selected, 3), row c * n_selected + j for copy c
 * of the j-th split-selected Gaussian (torch.normal(mean=0, std) / std of :264-266).
 * max_screen_size: the reference's `if max_screen_size:` (:316); 0 = None.  When set, Gaussians
 * whose largest world-space scale exceeds 0.1 * extent are pruned (big_points_ws, :318); the
 * screen-size half (max_radii2D > max_screen_size, :317) can never fire -- densification_postfix
 * has just zeroed max_radii2D (:249-251) -- so max_radii2D is not an input.  When 0, only the
 * opacity test (and NaN scales, :328-329) prunes.
 *
 * mgr_adam_step_groups: the same update with one step count per group (steps[k] <= 0 skips
 * group k): torch.optim.Adam keeps state["step"] per parameter, and the reference replaces
 * leaves (reset_opacity / densify / prune run in on_after_backward, hand_dynamic.py:193-224)
 * before optimizer.step(), which then skips the gradient-less new nn.Parameter.
 *
 * mgr_prune_plan: prune_points(mask), gaussian.py:185-203 (the mask_to_prune branch of
 * density_update, gaussian_utils.py:454-459): prune_mask (N) bytes, non-zero = remove.  Builds the
 * same source map as mgr_densify_plan (counts_host[0] = counts_host[4] = survivors); the rows are
 * then written by mgr_densify_apply(N, M, 0, ..., noise = NULL).
 *
 * mgr_gather_rows: dst[o,:] = src[map[o],:] for the M rows of the plan in `workspace`, rows of
 * `width` 4-byte words: the per-Gaussian statistics (xyz_gradient_accum, denom, max_radii2D,
 * gaussian.py:199-203) and any other per-Gaussian side array.
 * ------------------------------------------------------------------------ */
int mgr_adam_step(int n_groups, const int64_t* counts, float* const* params, const float* const* grads,
                  float* const* exp_avg, float* const* exp_avg_sq, const double* lrs, int64_t step, double beta1,
                  double beta2, double eps, void* stream);
int mgr_adam_step_groups(int n_groups, const int64_t* counts, float* const* params, const float* const* grads,
                         float* const* exp_avg, float* const* exp_avg_sq, const double* lrs, const int64_t* steps,
                         double beta1, double beta2, double eps, void* stream);
int mgr_reset_opacity(int N, float* opacity_logit, float* exp_avg, float* exp_avg_sq, void* stream);
/* add_densification_stats, gaussian.py:335-338 (xyz_gradient_accum += the step's sum over views of ||dL/dmeans2D[:, :2]||,
 * denom += the number of views that saw the Gaussian) and the max_radii2D update of density_update,
 * gaussian_utils.py:470-473 (max_radii2D = max(max_radii2D, radii)), in one launch.  grad2d_sum / vis_count: (N,) fp32,
 * radii_max: (N,) int32 -- the statistics outputs of mgr_views_backward; the three accumulators (N,) fp32, in place. */
int mgr_add_densification_stats(int N, const float* grad2d_sum, const float* vis_count, const int32_t* radii_max,
                                float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream);
size_t mgr_densify_workspace_bytes(int N);
int mgr_densify_plan(int N, const float* grad_accum, const float* denom, const float* log_scale,
                     const float* opacity_logit, float max_grad, float min_opacity, float extent, float percent_dense,
                     float max_screen_size, void* workspace, size_t workspace_bytes, int64_t* counts_host,
                     void* stream);
int mgr_prune_plan(int N, const uint8_t* prune_mask, void* workspace, size_t workspace_bytes, int64_t* counts_host,
                   void* stream);
int mgr_gather_rows(int N, int64_t M, const void* workspace, const void* src, void* dst, int width, void* stream);
int mgr_densify_apply(int N, int64_t M, int64_t n_selected, const void* workspace, const float* const* params,
                      const float* const* exp_avg, const float* const* exp_avg_sq, float* const* new_params,
                      float* const* new_exp_avg, float* const* new_exp_avg_sq, const float* skin, float* new_skin, int B,
                      const float* noise, void* stream);

/* Isotropic regulariser of loss_func (src/modules/base.py:349-356):
 *   loss[0] = weight * mean_n (min_j s_nj / (max_j s_nj + 1e-8) - condition_number)^2, s = exp(log_scale (N,3));
 *   d_log_scale (N,3) = its gradient (added to the existing content when accumulate != 0).
 * workspace: mgr_isotropic_reg_workspace_bytes(N). */
size_t mgr_isotropic_reg_workspace_bytes(int N);
int mgr_isotropic_reg(int N, const float* log_scale, float condition_number, float weight, float* d_log_scale,
                      int accumulate, float* loss, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Contact distance: for each of the N1 points of pt1
 * (N1,3) the fp32 distance to its nearest point of pt2 (N2,3) and that point's
 * index.  Replaces get_contact_dist (taichi kernel, src/utils/gaussian_utils.py:
 * 521-549) and get_contact_map (chunked torch.cdist().min(), :514-518).  Same
 * loop semantics: dist = sqrt(dx^2+dy^2+dz^2), strict '<' on the rooted
 * distance (lowest index wins a tie), initial minimum 1e9 (N2 = 0 -> 1e9, index 0).
 * out_idx (int32) may be NULL.
 * ------------------------------------------------------------------------ */
size_t mgr_contact_workspace_bytes(int N1, int N2);
int mgr_contact_dist(int N1, const float* pt1, int N2, const float* pt2, float* out_dist, int32_t* out_idx,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Skin-weight initialisation from the MANO rest mesh (model-initialisation side of the
 * dataloader): the device half of init_mano_weights (src/utils/train_utils.py:48-89) as called by
 * Dataset.build_voxel_grid / Dataset.sample_gaussians_on_bones (src/datasets/brics_dynamic.py:69-144).
 *
 * mgr_knn_mean_rows: for each of the n points (n,3) its k nearest of the m reference points (m,3) by squared Euclidean
 *   distance (replaces torch.cdist(points, mano_verts).topk(k, largest=False), train_utils.py:70-72; ties keep the lower
 *   index), out (n,C) = mean of rows[idx] (m,C) added nearest first in fp32 (np.mean(init_weights[indices], axis=1),
 *   :73), out_idx (n,k) int32 the indices (nearest first; -1 beyond m).  Either output may be NULL.  1 <= k <= 32.
 * mgr_mesh_sdf: signed distance (n,) of the points to the triangle mesh verts (nv,3) / faces (nf,3) int32, positive
 *   inside (replaces pysdf.SDF(verts, faces)(points), train_utils.py:55-58; pysdf is an external package that is not in
 *   this image: parity unpinned).  |distance| is the exact point-triangle minimum; the sign comes from the generalised
 *   winding number (|sum of solid angles| / 4 pi > 1/2), also written to out_winding (n,) when not NULL.
 * ------------------------------------------------------------------------ */
int mgr_knn_mean_rows(int n, const float* points, int m, const float* refs, const float* rows, int C, int k, float* out,
                      int32_t* out_idx, void* stream);
int mgr_mesh_sdf(int n, const float* points, int nv, const float* verts, int nf, const int32_t* faces, float* out_sdf,
                 float* out_winding, void* stream);

/* ------------------------------------------------------------------------
 * Row-compacted gradient exchange of the view-sharded step (no reference counterpart: the reference trains
 * on one GPU, /root/reference/main.py:84-87).  The step buffer `flat` holds nseg segments of N rows (segment k: widths[k]
 * floats per row, first float at offs[k]), the visibility counts (N floats at vis_off) and (loss, overflow) at tail_off.
 * Between two SUM all-reduces issued by the host (torch.distributed / RCCL):
 *   mgr_exchange_mask    small[0:N] = 1 where the row may be non-zero -- from the rows themselves, or (active_list /
 *                        active_count: device pointers of mgr_views_active_list) from the fused backward's list of the
 *                        Gaussians that received a gradient --, small[N:2N] = the visibility counts as bytes
 *   mgr_exchange_index   idx = the rows with mask != 0 in ascending order, *count (device) their number
 *   mgr_exchange_pack    buf = [segment 0 rows (n x widths[0]) | segment 1 rows | ... | loss, overflow]
 *   mgr_exchange_unpack  the inverse into `flat` (rows outside idx untouched) and, when small_vis is given, the visibility
 *                        counts back as floats
 * ------------------------------------------------------------------------ */
int mgr_exchange_mask(int N, const float* flat, int nseg, const int64_t* offs, const int* widths, int64_t vis_off,
                      const uint32_t* active_list, const uint32_t* active_count, uint8_t* small, void* stream);
size_t mgr_exchange_index_workspace_bytes(int N);
int mgr_exchange_index(int N, const uint8_t* mask, uint32_t* idx, uint32_t* count, void* workspace, size_t workspace_bytes,
                       void* stream);
int mgr_exchange_pack(int N, int n, const uint32_t* idx, const float* flat, int nseg, const int64_t* offs, const int* widths,
                      int64_t tail_off, float* buf, void* stream);
int mgr_exchange_unpack(int N, int n, const uint32_t* idx, float* flat, int nseg, const int64_t* offs, const int* widths,
                        int64_t tail_off, const float* buf, const uint8_t* small_vis, int64_t vis_off, void* stream);
/* The same two with the row count read FROM THE DEVICE (`count`: the word mgr_exchange_index wrote) and a row capacity
 * `cap_rows` <= N chosen by the host beforehand -- the second collective (cap_rows x columns + 2 floats, segment k at
 * cap_rows x its first column) is sized without a host read in the middle of the step.  Rows [min(*count, cap_rows), cap_rows)
 * are packed as zeros and ignored by the unpack; *count > cap_rows adds 1 to the packed overflow word (buf's last float), which
 * the all-reduce sums and the unpack writes back: the caller runs the step again with a larger capacity. */
int mgr_exchange_pack_rows(int N, int cap_rows, const uint32_t* count, const uint32_t* idx, const float* flat, int nseg, const int64_t* offs,
                           const int* widths, int64_t tail_off, float* buf, void* stream);
int mgr_exchange_unpack_rows(int N, int cap_rows, const uint32_t* count, const uint32_t* idx, float* flat, int nseg, const int64_t* offs,
                             const int* widths, int64_t tail_off, const float* buf, const uint8_t* small_vis, int64_t vis_off, void* stream);

/* ------------------------------------------------------------------------
 * Measurement aid: when enabled, every kernel launched by this library is
 * bracketed by HIP events recorded on the caller's stream.
 * mgr_profile_report synchronises the stream, writes one line per kernel
 * "name count total_ms\n" into buf (NUL terminated) and clears the records.
 * ------------------------------------------------------------------------ */
int mgr_profile_enable(int on);
/* Bracket only the kernel of this name (as it appears in the report); NULL or "" = every kernel.  Two events per
 * bracketed launch cost a few microseconds of GPU time each: a run that is itself being timed should name the one
 * kernel it needs. */
int mgr_profile_filter(const char* kernel_name);
/* Bracket only every n-th launch that passes the filter (n <= 1: every one): the two events around a launch keep the
 * kernels on either side from following it without a gap (~6 us each on this GPU), so a run that is being timed samples. */
int mgr_profile_sample_every(int n);
int mgr_profile_report(char* buf_host, size_t len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MANUS_HIP_H */
