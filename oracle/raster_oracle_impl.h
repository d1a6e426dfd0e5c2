/* CPU ORACLE — test infrastructure only, never linked into the product.
 *
 * Scalar restatement of the external tile rasterizer that MANUS calls at
 * /root/reference/src/utils/gaussian_utils.py:393-416 (`GaussianRasterizer`,
 * cloned unpinned by /root/reference/setup_env.sh:6 from
 * graphdeco-inria/diff-gaussian-rasterization; the source is NOT in the
 * reference tree, so this follows the published algorithm as restated in
 * SURVEY.md Appendix A).  PARITY UNPINNED: the reference ships no tests or
 * golden vectors for this boundary and the CUDA binary cannot run here; this
 * oracle is validated instead by analytic cases, invariants and an fp64
 * finite-difference check (tests/test_oracle_raster.py).
 *
 * This header is included twice by raster_oracle.c, once with REAL=float
 * (suffix _f32: same precision as the product) and once with REAL=double
 * (suffix _f64: the "true" value used for gradient tolerances).
 *
 * Only the colors_precomp + cov3D_precomp variant is implemented — the only
 * one MANUS exercises (gaussian_utils.py:407-416, shs=None, scales=None).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)
#define R(x) ((REAL)(x))

typedef struct {
    int tile;
    REAL depth;
    int gid;
} FN(OrcPair);

typedef struct FN(OrcState) {
    int W, H, N, gx, gy;
    REAL tanfovx, tanfovy, view[16], proj[16], bg[3];
    /* inputs kept for backward */
    REAL *means3D, *cov3D, *colors;
    /* geometry state */
    REAL *xy, *depth, *conic_opacity;
    int *radii, *tiles_touched, *rect; /* rect: x0,y0,x1,y1 */
    /* binning state */
    long num_rendered;
    FN(OrcPair) * pairs;
    int *ranges; /* tiles x 2 */
    /* image state */
    REAL *final_T;
    int *n_contrib;
    /* test aid: forced outcomes of the alpha >= 1/255 test for listed (pixel, Gaussian) pairs, sorted by (pixel, gid);
     * empty in every parity run that is not closing a threshold-flip argument (orc_set_overrides) */
    int n_over;
    int *over_pix, *over_gid, *over_keep;
    /* test aid: per pixel the Gaussian the walk has to end on as its last contributor (-1 = no contributor; array NULL =
     * own rule); a pixel whose own T < 1e-4 decision differs is counted in stop_flips, and in stop_violations when its
     * test value was NOT within rounding of 1e-4 */
    int* forced_last;
    long stop_flips, stop_violations;
} FN(OrcState);

/* -1 = no override for this pair, else 0 (drop) / 1 (keep) */
static int FN(orc_override)(const FN(OrcState) * s, int pix, int gid) {
    int lo = 0, hi = s->n_over - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (s->over_pix[mid] < pix || (s->over_pix[mid] == pix && s->over_gid[mid] < gid)) lo = mid + 1;
        else if (s->over_pix[mid] == pix && s->over_gid[mid] == gid) return s->over_keep[mid];
        else hi = mid - 1;
    }
    return -1;
}

static REAL FN(orc_exp)(REAL x) { return (sizeof(REAL) == 4) ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static REAL FN(orc_sqrt)(REAL x) { return (sizeof(REAL) == 4) ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static REAL FN(orc_ceil)(REAL x) { return (sizeof(REAL) == 4) ? (REAL)ceilf((float)x) : (REAL)ceil((double)x); }
static REAL FN(orc_max)(REAL a, REAL b) { return a > b ? a : b; }
static REAL FN(orc_min)(REAL a, REAL b) { return a < b ? a : b; }
static int FN(orc_imin)(int a, int b) { return a < b ? a : b; }
static int FN(orc_imax)(int a, int b) { return a > b ? a : b; }

/* column-major 4x4 applied to a point: element (row r, col c) = M[4c+r]
 * (layout contract: /root/reference/src/utils/cam_utils.py:58-63). */
static void FN(orc_xform43)(const REAL* M, const REAL* p, REAL* o) {
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
static void FN(orc_xform44)(const REAL* M, const REAL* p, REAL* o) {
    FN(orc_xform43)(M, p, o);
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

/* EWA: rows M0, M1 of (J * Rwv), J the clamped perspective Jacobian.
 * Returns t (clamped view-space point) and the clamp masks. */
static void FN(orc_ewa_rows)(const FN(OrcState) * s, const REAL* mean, REAL* M0, REAL* M1, REAL* t,
                             REAL* xmul, REAL* ymul, REAL* fx_out, REAL* fy_out) {
    const REAL* v = s->view;
    REAL fx = R(s->W) / (R(2) * s->tanfovx), fy = R(s->H) / (R(2) * s->tanfovy);
    FN(orc_xform43)(v, mean, t);
    REAL limx = R(1.3) * s->tanfovx, limy = R(1.3) * s->tanfovy;
    REAL txtz = t[0] / t[2], tytz = t[1] / t[2];
    *xmul = (txtz < -limx || txtz > limx) ? R(0) : R(1);
    *ymul = (tytz < -limy || tytz > limy) ? R(0) : R(1);
    t[0] = FN(orc_min)(limx, FN(orc_max)(-limx, txtz)) * t[2];
    t[1] = FN(orc_min)(limy, FN(orc_max)(-limy, tytz)) * t[2];
    REAL j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
    REAL j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
    for (int c = 0; c < 3; ++c) { /* Rwv[r][c] = v[4c+r] */
        M0[c] = j00 * v[4 * c + 0] + j02 * v[4 * c + 2];
        M1[c] = j11 * v[4 * c + 1] + j12 * v[4 * c + 2];
    }
    *fx_out = fx;
    *fy_out = fy;
}

static void FN(orc_sym_mul)(const REAL* c6, const REAL* m, REAL* o) {
    /* o = Sigma * m, Sigma packed [xx,xy,xz,yy,yz,zz] */
    o[0] = c6[0] * m[0] + c6[1] * m[1] + c6[2] * m[2];
    o[1] = c6[1] * m[0] + c6[3] * m[1] + c6[4] * m[2];
    o[2] = c6[2] * m[0] + c6[4] * m[1] + c6[5] * m[2];
}

static int FN(orc_pair_cmp)(const void* a, const void* b) {
    const FN(OrcPair)* x = (const FN(OrcPair)*)a;
    const FN(OrcPair)* y = (const FN(OrcPair)*)b;
    if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
    if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
    /* equal (tile, depth bits): the upstream radix sort is stable and pairs are
     * emitted in Gaussian-index order, so ties keep index order */
    return x->gid < y->gid ? -1 : (x->gid > y->gid ? 1 : 0);
}

void FN(orc_free)(FN(OrcState) * s) {
    if (!s) return;
    free(s->means3D); free(s->cov3D); free(s->colors); free(s->xy); free(s->depth);
    free(s->conic_opacity); free(s->radii); free(s->tiles_touched); free(s->rect);
    free(s->pairs); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s->over_pix); free(s->over_gid); free(s->over_keep); free(s->forced_last);
    free(s);
}

/* K2-K6 from the per-Gaussian 2D state in `s` (xy, depth, conic_opacity, radii, rect, tiles_touched): emit and sort
 * the (tile, depth) pairs, tile ranges, front-to-back compositing.  Shared by orc_forward (state from its own
 * preprocess) and orc_forward_geom (state handed in). */
static void FN(orc_blend_forward)(FN(OrcState) * s, const REAL* colors, const REAL* bg, REAL* out_color);
static void FN(orc_bin_and_blend)(FN(OrcState) * s, const REAL* colors, const REAL* bg, REAL* out_color) {
    const int N = s->N;
    /* ---- K2-K5: emit (tile, depth) pairs, sort, tile ranges ----------- */
    long total = 0;
    for (int i = 0; i < N; ++i) total += s->tiles_touched[i];
    s->num_rendered = total;
    s->pairs = (FN(OrcPair)*)malloc((size_t)(total > 0 ? total : 1) * sizeof(FN(OrcPair)));
    long off = 0;
    for (int i = 0; i < N; ++i) {
        if (s->radii[i] <= 0) continue;
        for (int y = s->rect[4 * i + 1]; y < s->rect[4 * i + 3]; ++y)
            for (int x = s->rect[4 * i]; x < s->rect[4 * i + 2]; ++x) {
                s->pairs[off].tile = y * s->gx + x;
                s->pairs[off].depth = s->depth[i];
                s->pairs[off].gid = i;
                ++off;
            }
    }
    qsort(s->pairs, (size_t)total, sizeof(FN(OrcPair)), FN(orc_pair_cmp));
    int ntiles = s->gx * s->gy;
    s->ranges = (int*)calloc((size_t)ntiles * 2, sizeof(int));
    for (long k = 0; k < total; ++k) {
        int t = s->pairs[k].tile;
        if (k == 0 || s->pairs[k - 1].tile != t) s->ranges[2 * t] = (int)k;
        if (k == total - 1 || s->pairs[k + 1].tile != t) s->ranges[2 * t + 1] = (int)k + 1;
    }

    FN(orc_blend_forward)(s, colors, bg, out_color);
}

/* ---- K6: front-to-back alpha compositing over the tile ranges in `s` ---- */
static void FN(orc_blend_forward)(FN(OrcState) * s, const REAL* colors, const REAL* bg, REAL* out_color) {
    const int W = s->W, H = s->H;
    if (!s->final_T) s->final_T = (REAL*)malloc((size_t)W * H * sizeof(REAL));
    if (!s->n_contrib) s->n_contrib = (int*)malloc((size_t)W * H * sizeof(int));
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int tile = (py / 16) * s->gx + (px / 16);
            int beg = s->ranges[2 * tile], end = s->ranges[2 * tile + 1];
            REAL T = R(1), C[3] = {0, 0, 0};
            int contributor = 0, last = 0;
            const int forcing = s->forced_last != NULL;
            const int forced_gid = forcing ? s->forced_last[py * W + px] : -1;
            int seen_last = forced_gid < 0, flipped = 0;   /* seen_last: the forced last contributor has been composited */
            for (int k = beg; k < end; ++k) {
                int g = s->pairs[k].gid;
                ++contributor;
                REAL dx = s->xy[2 * g] - R(px), dy = s->xy[2 * g + 1] - R(py);
                const REAL* co = s->conic_opacity + 4 * g;
                REAL power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > R(0)) continue;
                REAL alpha = FN(orc_min)(R(0.99), co[3] * FN(orc_exp)(power));
                int keep = alpha >= R(1) / R(255);
                if (s->n_over) {
                    int ov = FN(orc_override)(s, py * W + px, g);
                    if (ov >= 0) keep = ov;
                }
                if (!keep) continue;
                REAL testT = T * (R(1) - alpha);
                int stop = testT < R(0.0001);
                if (forcing) {
                    const int fstop = seen_last;
                    if (g == forced_gid) seen_last = 1;
                    if (fstop != stop && !flipped) {
                        flipped = 1;
                        ++s->stop_flips;
                        REAL dev = testT / R(0.0001) - R(1);
                        if (dev > R(1e-4) || dev < R(-1e-4)) ++s->stop_violations;
                    }
                    stop = fstop;
                }
                if (stop) break;
                for (int ch = 0; ch < 3; ++ch) C[ch] += colors[3 * g + ch] * alpha * T;
                T = testT;
                last = contributor;
            }
            size_t pix = (size_t)py * W + px;
            s->final_T[pix] = T;
            s->n_contrib[pix] = last;
            for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
        }
}

/* Forward.  out_color is CHW (3,H,W); radii int32 (N).  do_blend = 0 stops after K1 (the per-Gaussian preprocess: radii,
 * depth, pixel centres, conics, tile rectangles, num_rendered): what a test of the integer state at full size needs. */
FN(OrcState) * FN(orc_forward_ex)(int W, int H, REAL tanfovx, REAL tanfovy, const REAL* view,
                               const REAL* proj, int N, const REAL* means3D, const REAL* cov3D,
                               const REAL* colors, const REAL* opacity, const REAL* bg,
                               REAL* out_color, int* radii_out, int do_blend) {
    FN(OrcState)* s = (FN(OrcState)*)calloc(1, sizeof(FN(OrcState)));
    s->W = W; s->H = H; s->N = N;
    s->gx = (W + 15) / 16; s->gy = (H + 15) / 16;
    s->tanfovx = tanfovx; s->tanfovy = tanfovy;
    memcpy(s->view, view, sizeof(REAL) * 16);
    memcpy(s->proj, proj, sizeof(REAL) * 16);
    memcpy(s->bg, bg, sizeof(REAL) * 3);
    size_t n = (size_t)(N > 0 ? N : 1);
    s->means3D = (REAL*)malloc(n * 3 * sizeof(REAL));
    s->cov3D = (REAL*)malloc(n * 6 * sizeof(REAL));
    s->colors = (REAL*)malloc(n * 3 * sizeof(REAL));
    memcpy(s->means3D, means3D, (size_t)N * 3 * sizeof(REAL));
    memcpy(s->cov3D, cov3D, (size_t)N * 6 * sizeof(REAL));
    memcpy(s->colors, colors, (size_t)N * 3 * sizeof(REAL));
    s->xy = (REAL*)calloc(n * 2, sizeof(REAL));
    s->depth = (REAL*)calloc(n, sizeof(REAL));
    s->conic_opacity = (REAL*)calloc(n * 4, sizeof(REAL));
    s->radii = (int*)calloc(n, sizeof(int));
    s->tiles_touched = (int*)calloc(n, sizeof(int));
    s->rect = (int*)calloc(n * 4, sizeof(int));

    /* ---- K1: per-Gaussian preprocess ---------------------------------- */
    long total = 0;
    for (int i = 0; i < N; ++i) {
        const REAL* p = means3D + 3 * i;
        REAL pv[3], ph[4];
        FN(orc_xform43)(view, p, pv);
        if (pv[2] <= R(0.2)) continue; /* near cull */
        FN(orc_xform44)(proj, p, ph);
        REAL pw = R(1) / (ph[3] + R(0.0000001));
        REAL px = ph[0] * pw, py = ph[1] * pw;
        REAL M0[3], M1[3], t[3], xm, ym, fx, fy, S0[3], S1[3];
        FN(orc_ewa_rows)(s, p, M0, M1, t, &xm, &ym, &fx, &fy);
        FN(orc_sym_mul)(cov3D + 6 * i, M0, S0);
        FN(orc_sym_mul)(cov3D + 6 * i, M1, S1);
        REAL a = M0[0] * S0[0] + M0[1] * S0[1] + M0[2] * S0[2] + R(0.3);
        REAL b = M0[0] * S1[0] + M0[1] * S1[1] + M0[2] * S1[2];
        REAL c = M1[0] * S1[0] + M1[1] * S1[1] + M1[2] * S1[2] + R(0.3);
        REAL det = a * c - b * b;
        if (det == R(0)) continue;
        REAL dinv = R(1) / det;
        REAL mid = R(0.5) * (a + c);
        REAL sq = FN(orc_sqrt)(FN(orc_max)(R(0.1), mid * mid - det));
        REAL l1 = mid + sq, l2 = mid - sq;
        int radius = (int)FN(orc_ceil)(R(3) * FN(orc_sqrt)(FN(orc_max)(l1, l2)));
        REAL ix = ((px + R(1)) * R(W) - R(1)) * R(0.5);
        REAL iy = ((py + R(1)) * R(H) - R(1)) * R(0.5);
        int x0 = FN(orc_imin)(s->gx, FN(orc_imax)(0, (int)((ix - R(radius)) / R(16))));
        int y0 = FN(orc_imin)(s->gy, FN(orc_imax)(0, (int)((iy - R(radius)) / R(16))));
        int x1 = FN(orc_imin)(s->gx, FN(orc_imax)(0, (int)((ix + R(radius) + R(15)) / R(16))));
        int y1 = FN(orc_imin)(s->gy, FN(orc_imax)(0, (int)((iy + R(radius) + R(15)) / R(16))));
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        s->depth[i] = pv[2];
        s->radii[i] = radius;
        s->xy[2 * i] = ix; s->xy[2 * i + 1] = iy;
        s->conic_opacity[4 * i + 0] = c * dinv;
        s->conic_opacity[4 * i + 1] = -b * dinv;
        s->conic_opacity[4 * i + 2] = a * dinv;
        s->conic_opacity[4 * i + 3] = opacity[i];
        s->tiles_touched[i] = (x1 - x0) * (y1 - y0);
        s->rect[4 * i] = x0; s->rect[4 * i + 1] = y0; s->rect[4 * i + 2] = x1; s->rect[4 * i + 3] = y1;
        total += s->tiles_touched[i];
    }
    if (radii_out) memcpy(radii_out, s->radii, (size_t)N * sizeof(int));
    if (!do_blend) {
        s->num_rendered = total;
        return s;
    }
    FN(orc_bin_and_blend)(s, colors, bg, out_color);
    return s;
}

FN(OrcState) * FN(orc_forward)(int W, int H, REAL tanfovx, REAL tanfovy, const REAL* view,
                               const REAL* proj, int N, const REAL* means3D, const REAL* cov3D,
                               const REAL* colors, const REAL* opacity, const REAL* bg,
                               REAL* out_color, int* radii_out) {
    return FN(orc_forward_ex)(W, H, tanfovx, tanfovy, view, proj, N, means3D, cov3D, colors, opacity, bg, out_color, radii_out, 1);
}

/* Forward from given 2D state: the blend (K2-K6) of SURVEY.md Appendix A on per-Gaussian screen-space inputs
 * xy (N,2) pixel centres, depth (N), conic_opacity (N,4) = (A, B, C, opacity), radii (N) (0 = culled), colours (N,3).
 * The tile rectangle is re-derived from (xy, radius) with the K1 rule.  Used to check a kernel that fuses its own
 * preprocess (LBS / SH / projection) against the oracle on IDENTICAL rasterizer inputs. */
FN(OrcState) * FN(orc_forward_geom)(int W, int H, int N, const REAL* xy, const REAL* depth, const REAL* conic_opacity,
                                    const int* radii, const REAL* colors, const REAL* bg, REAL* out_color) {
    FN(OrcState)* s = (FN(OrcState)*)calloc(1, sizeof(FN(OrcState)));
    s->W = W; s->H = H; s->N = N;
    s->gx = (W + 15) / 16; s->gy = (H + 15) / 16;
    memcpy(s->bg, bg, sizeof(REAL) * 3);
    size_t n = (size_t)(N > 0 ? N : 1);
    s->colors = (REAL*)malloc(n * 3 * sizeof(REAL));
    memcpy(s->colors, colors, (size_t)N * 3 * sizeof(REAL));
    s->xy = (REAL*)calloc(n * 2, sizeof(REAL));
    s->depth = (REAL*)calloc(n, sizeof(REAL));
    s->conic_opacity = (REAL*)calloc(n * 4, sizeof(REAL));
    s->radii = (int*)calloc(n, sizeof(int));
    s->tiles_touched = (int*)calloc(n, sizeof(int));
    s->rect = (int*)calloc(n * 4, sizeof(int));
    for (int i = 0; i < N; ++i) {
        int radius = radii[i];
        if (radius <= 0) continue;
        REAL ix = xy[2 * i], iy = xy[2 * i + 1];
        int x0 = FN(orc_imin)(s->gx, FN(orc_imax)(0, (int)((ix - R(radius)) / R(16))));
        int y0 = FN(orc_imin)(s->gy, FN(orc_imax)(0, (int)((iy - R(radius)) / R(16))));
        int x1 = FN(orc_imin)(s->gx, FN(orc_imax)(0, (int)((ix + R(radius) + R(15)) / R(16))));
        int y1 = FN(orc_imin)(s->gy, FN(orc_imax)(0, (int)((iy + R(radius) + R(15)) / R(16))));
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        s->depth[i] = depth[i];
        s->radii[i] = radius;
        s->xy[2 * i] = ix; s->xy[2 * i + 1] = iy;
        memcpy(s->conic_opacity + 4 * i, conic_opacity + 4 * i, 4 * sizeof(REAL));
        s->tiles_touched[i] = (x1 - x0) * (y1 - y0);
        s->rect[4 * i] = x0; s->rect[4 * i + 1] = y0; s->rect[4 * i + 2] = x1; s->rect[4 * i + 3] = y1;
    }
    FN(orc_bin_and_blend)(s, colors, bg, out_color);
    return s;
}

long FN(orc_num_rendered)(const FN(OrcState) * s) { return s->num_rendered; }

/* ---- test aids for the threshold-flip argument (tests/test_gpu_flips.py) --------------------------------------
 * The kernels evaluate exp through v_exp_f32 in the log2 domain, this oracle through expf: on bit-identical inputs
 * the two can disagree on alpha >= 1/255 only for pairs whose alpha lies within rounding of the threshold.
 * orc_collect_ambiguous lists the (pixel, Gaussian) pairs the forward walk evaluated with |alpha * 255 - 1| <= eps;
 * orc_reblend_with_overrides forces the outcome of the test for given pairs (sorted by (pixel, gid)) and composites
 * again -- forward state, image and the following backward then follow the forced decisions.  The same holds for the
 * second threshold, T (1 - alpha) < 1e-4: with `forced_last` (the kernels' n_contrib) every pixel's walk ends on the
 * given contributor, and the pixels whose own decision differed are counted -- together with those among them whose
 * test value was NOT within 1e-4 (relative) of the threshold, which must be none. */
long FN(orc_collect_ambiguous)(const FN(OrcState) * s, REAL eps, long cap, int* pix_out, int* gid_out, REAL* alpha_out) {
    const int W = s->W, H = s->H;
    long n = 0;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int tile = (py / 16) * s->gx + (px / 16);
            int beg = s->ranges[2 * tile], end = s->ranges[2 * tile + 1];
            REAL T = R(1);
            for (int k = beg; k < end; ++k) {
                int g = s->pairs[k].gid;
                REAL dx = s->xy[2 * g] - R(px), dy = s->xy[2 * g + 1] - R(py);
                const REAL* co = s->conic_opacity + 4 * g;
                REAL power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > R(0)) continue;
                REAL alpha = FN(orc_min)(R(0.99), co[3] * FN(orc_exp)(power));
                REAL d = alpha * R(255) - R(1);
                if (d <= eps && d >= -eps) {
                    if (n < cap) { pix_out[n] = py * W + px; gid_out[n] = g; alpha_out[n] = alpha; }
                    ++n;
                }
                if (alpha < R(1) / R(255)) continue;
                REAL testT = T * (R(1) - alpha);
                if (testT < R(0.0001)) break;
                T = testT;
            }
        }
    return n;
}

void FN(orc_reblend_with_overrides)(FN(OrcState) * s, int n, const int* pix, const int* gid, const int* keep,
                                    const int* forced_last, const REAL* colors, const REAL* bg, REAL* out_color,
                                    long* stop_flips, long* stop_violations) {
    free(s->forced_last);
    s->forced_last = NULL;
    if (forced_last) {
        s->forced_last = (int*)malloc((size_t)s->W * s->H * sizeof(int));
        memcpy(s->forced_last, forced_last, (size_t)s->W * s->H * sizeof(int));
    }
    s->stop_flips = s->stop_violations = 0;
    free(s->over_pix); free(s->over_gid); free(s->over_keep);
    s->n_over = n;
    s->over_pix = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    s->over_gid = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    s->over_keep = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    memcpy(s->over_pix, pix, (size_t)n * sizeof(int));
    memcpy(s->over_gid, gid, (size_t)n * sizeof(int));
    memcpy(s->over_keep, keep, (size_t)n * sizeof(int));
    FN(orc_blend_forward)(s, colors, bg, out_color);
    if (stop_flips) *stop_flips = s->stop_flips;
    if (stop_violations) *stop_violations = s->stop_violations;
}

void FN(orc_get_geom)(const FN(OrcState) * s, REAL* xy, REAL* depth, REAL* conic_opacity,
                      int* tiles_touched, int* rect) {
    if (xy) memcpy(xy, s->xy, (size_t)s->N * 2 * sizeof(REAL));
    if (depth) memcpy(depth, s->depth, (size_t)s->N * sizeof(REAL));
    if (conic_opacity) memcpy(conic_opacity, s->conic_opacity, (size_t)s->N * 4 * sizeof(REAL));
    if (tiles_touched) memcpy(tiles_touched, s->tiles_touched, (size_t)s->N * sizeof(int));
    if (rect) memcpy(rect, s->rect, (size_t)s->N * 4 * sizeof(int));
}

void FN(orc_get_binning)(const FN(OrcState) * s, int* point_list, int* ranges) {
    if (point_list)
        for (long k = 0; k < s->num_rendered; ++k) point_list[k] = s->pairs[k].gid;
    if (ranges) memcpy(ranges, s->ranges, (size_t)s->gx * s->gy * 2 * sizeof(int));
}

void FN(orc_get_image_state)(const FN(OrcState) * s, REAL* final_T, int* n_contrib) {
    if (final_T) memcpy(final_T, s->final_T, (size_t)s->W * s->H * sizeof(REAL));
    if (n_contrib) memcpy(n_contrib, s->n_contrib, (size_t)s->W * s->H * sizeof(int));
}

/* K7: per pixel, back to front.  Accumulates (outputs must be zeroed): dL_dmeans2D (N,3; NDC-scaled pixels, z = 0),
 * dconic (N,3) = d/d(A,B,C), dL_dcolors (N,3), dL_dopacity (N). */
static void FN(orc_blend_backward)(const FN(OrcState) * s, const REAL* dL_dpix, REAL* dL_dmeans2D, REAL* dconic,
                                   REAL* dL_dcolors, REAL* dL_dopacity) {
    const int W = s->W, H = s->H;
    /* ---- K7: back-to-front per pixel ----------------------------------- */
    REAL ddelx = R(0.5) * R(W), ddely = R(0.5) * R(H);
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            size_t pix = (size_t)py * W + px;
            int tile = (py / 16) * s->gx + (px / 16);
            int beg = s->ranges[2 * tile];
            REAL Tf = s->final_T[pix], T = Tf;
            int last = s->n_contrib[pix];
            REAL g[3], accum[3] = {0, 0, 0}, lastc[3] = {0, 0, 0}, last_alpha = 0;
            for (int ch = 0; ch < 3; ++ch) g[ch] = dL_dpix[(size_t)ch * H * W + pix];
            REAL bgdot = s->bg[0] * g[0] + s->bg[1] * g[1] + s->bg[2] * g[2];
            for (int k = beg + last - 1; k >= beg; --k) {
                int gi = s->pairs[k].gid;
                REAL dx = s->xy[2 * gi] - R(px), dy = s->xy[2 * gi + 1] - R(py);
                const REAL* co = s->conic_opacity + 4 * gi;
                REAL power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > R(0)) continue;
                REAL G = FN(orc_exp)(power);
                REAL alpha = FN(orc_min)(R(0.99), co[3] * G);
                int keep = alpha >= R(1) / R(255);
                if (s->n_over) {
                    int ov = FN(orc_override)(s, py * W + px, gi);
                    if (ov >= 0) keep = ov;
                }
                if (!keep) continue;
                T = T / (R(1) - alpha);
                REAL dch = alpha * T, dalpha = 0;
                for (int ch = 0; ch < 3; ++ch) {
                    REAL c = s->colors[3 * gi + ch];
                    accum[ch] = last_alpha * lastc[ch] + (R(1) - last_alpha) * accum[ch];
                    lastc[ch] = c;
                    dalpha += (c - accum[ch]) * g[ch];
                    dL_dcolors[3 * gi + ch] += dch * g[ch];
                }
                dalpha *= T;
                last_alpha = alpha;
                dalpha += (-Tf / (R(1) - alpha)) * bgdot;
                REAL dG = co[3] * dalpha;
                REAL gdx = G * dx, gdy = G * dy;
                REAL dGdx = -gdx * co[0] - gdy * co[1];
                REAL dGdy = -gdy * co[2] - gdx * co[1];
                dL_dmeans2D[3 * gi + 0] += dG * dGdx * ddelx;
                dL_dmeans2D[3 * gi + 1] += dG * dGdy * ddely;
                dconic[3 * gi + 0] += R(-0.5) * gdx * dx * dG;
                dconic[3 * gi + 1] += R(-0.5) * gdx * dy * dG;
                dconic[3 * gi + 2] += R(-0.5) * gdy * dy * dG;
                dL_dopacity[gi] += G * dalpha;
            }
        }

}

/* Backward of orc_forward_geom: the blend backward alone (outputs as in orc_blend_backward, fully written). */
void FN(orc_backward_geom)(const FN(OrcState) * s, const REAL* dL_dpix, REAL* dL_dmeans2D, REAL* dL_dconic,
                           REAL* dL_dcolors, REAL* dL_dopacity) {
    int N = s->N;
    memset(dL_dmeans2D, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dconic, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dcolors, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dopacity, 0, (size_t)N * sizeof(REAL));
    FN(orc_blend_backward)(s, dL_dpix, dL_dmeans2D, dL_dconic, dL_dcolors, dL_dopacity);
}

/* Backward.  dL_dpix CHW.  Outputs: dL_dmeans3D (N,3), dL_dmeans2D (N,3; z=0),
 * dL_dcolors (N,3), dL_dopacity (N), dL_dcov3D (N,6); optional dL_dconic (N,3)
 * = the per-Gaussian (A,B,C) conic gradient before the Sigma2D chain. */
void FN(orc_backward)(const FN(OrcState) * s, const REAL* dL_dpix, REAL* dL_dmeans3D,
                      REAL* dL_dmeans2D, REAL* dL_dcolors, REAL* dL_dopacity, REAL* dL_dcov3D,
                      REAL* dL_dconic_out) {
    int N = s->N;
    size_t n = (size_t)(N > 0 ? N : 1);
    REAL* dconic = (REAL*)calloc(n * 3, sizeof(REAL));
    memset(dL_dmeans3D, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dmeans2D, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dcolors, 0, (size_t)N * 3 * sizeof(REAL));
    memset(dL_dopacity, 0, (size_t)N * sizeof(REAL));
    memset(dL_dcov3D, 0, (size_t)N * 6 * sizeof(REAL));

    FN(orc_blend_backward)(s, dL_dpix, dL_dmeans2D, dconic, dL_dcolors, dL_dopacity);

    /* ---- K8 + K9: per-Gaussian preprocess backward --------------------- */
    for (int i = 0; i < N; ++i) {
        if (!(s->radii[i] > 0)) continue;
        const REAL* p = s->means3D + 3 * i;
        const REAL* c6 = s->cov3D + 6 * i;
        const REAL* v = s->view;
        REAL M0[3], M1[3], t[3], xm, ym, fx, fy, S0[3], S1[3];
        FN(orc_ewa_rows)(s, p, M0, M1, t, &xm, &ym, &fx, &fy);
        FN(orc_sym_mul)(c6, M0, S0);
        FN(orc_sym_mul)(c6, M1, S1);
        REAL a = M0[0] * S0[0] + M0[1] * S0[1] + M0[2] * S0[2] + R(0.3);
        REAL b = M0[0] * S1[0] + M0[1] * S1[1] + M0[2] * S1[2];
        REAL c = M1[0] * S1[0] + M1[1] * S1[1] + M1[2] * S1[2] + R(0.3);
        REAL dA = dconic[3 * i], dB = dconic[3 * i + 1], dC = dconic[3 * i + 2];
        REAL den = a * c - b * b;
        REAL k = R(1) / (den * den + R(0.0000001));
        REAL da = 0, db = 0, dc = 0;
        if (k != R(0)) {
            da = k * (-c * c * dA + R(2) * b * c * dB + (den - a * c) * dC);
            dc = k * (-a * a * dC + R(2) * a * b * dB + (den - a * c) * dA);
            db = k * R(2) * (b * c * dA - (den + R(2) * b * b) * dB + a * b * dC);
            REAL* o = dL_dcov3D + 6 * i;
            o[0] = M0[0] * M0[0] * da + M0[0] * M1[0] * db + M1[0] * M1[0] * dc;
            o[3] = M0[1] * M0[1] * da + M0[1] * M1[1] * db + M1[1] * M1[1] * dc;
            o[5] = M0[2] * M0[2] * da + M0[2] * M1[2] * db + M1[2] * M1[2] * dc;
            o[1] = R(2) * M0[0] * M0[1] * da + (M0[0] * M1[1] + M0[1] * M1[0]) * db + R(2) * M1[0] * M1[1] * dc;
            o[2] = R(2) * M0[0] * M0[2] * da + (M0[0] * M1[2] + M0[2] * M1[0]) * db + R(2) * M1[0] * M1[2] * dc;
            o[4] = R(2) * M0[2] * M0[1] * da + (M0[1] * M1[2] + M0[2] * M1[1]) * db + R(2) * M1[1] * M1[2] * dc;
        }
        REAL dM0[3], dM1[3];
        for (int j = 0; j < 3; ++j) {
            dM0[j] = R(2) * S0[j] * da + S1[j] * db;
            dM1[j] = R(2) * S1[j] * dc + S0[j] * db;
        }
        /* M0[c] = j00*R[0][c] + j02*R[2][c];  R[r][c] = v[4c+r] */
        REAL dJ00 = v[0] * dM0[0] + v[4] * dM0[1] + v[8] * dM0[2];
        REAL dJ02 = v[2] * dM0[0] + v[6] * dM0[1] + v[10] * dM0[2];
        REAL dJ11 = v[1] * dM1[0] + v[5] * dM1[1] + v[9] * dM1[2];
        REAL dJ12 = v[2] * dM1[0] + v[6] * dM1[1] + v[10] * dM1[2];
        REAL tz = R(1) / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        REAL dtx = xm * -fx * tz2 * dJ02;
        REAL dty = ym * -fy * tz2 * dJ12;
        REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (R(2) * fx * t[0]) * tz3 * dJ02 +
                   (R(2) * fy * t[1]) * tz3 * dJ12;
        REAL dm[3];
        dm[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;
        dm[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
        dm[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;
        /* projective part: mean2D (NDC-scaled pixels) -> mean3D */
        const REAL* P = s->proj;
        REAL hw = P[3] * p[0] + P[7] * p[1] + P[11] * p[2] + P[15];
        REAL mw = R(1) / (hw + R(0.0000001));
        REAL mul1 = (P[0] * p[0] + P[4] * p[1] + P[8] * p[2] + P[12]) * mw * mw;
        REAL mul2 = (P[1] * p[0] + P[5] * p[1] + P[9] * p[2] + P[13]) * mw * mw;
        REAL gx2 = dL_dmeans2D[3 * i], gy2 = dL_dmeans2D[3 * i + 1];
        dm[0] += (P[0] * mw - P[3] * mul1) * gx2 + (P[1] * mw - P[3] * mul2) * gy2;
        dm[1] += (P[4] * mw - P[7] * mul1) * gx2 + (P[5] * mw - P[7] * mul2) * gy2;
        dm[2] += (P[8] * mw - P[11] * mul1) * gx2 + (P[9] * mw - P[11] * mul2) * gy2;
        dL_dmeans3D[3 * i] = dm[0];
        dL_dmeans3D[3 * i + 1] = dm[1];
        dL_dmeans3D[3 * i + 2] = dm[2];
    }
    if (dL_dconic_out) memcpy(dL_dconic_out, dconic, (size_t)N * 3 * sizeof(REAL));
    free(dconic);
}

#undef CAT_
#undef CAT
#undef FN
#undef R
