// hostcheck.hip -- TEST SHIM: runs the product's __host__ __device__ per-Gaussian / per-pixel math
// (reduced-3dgs_amd/csrc/gauss_math.h, blend_math.h) on the CPU, so tests/test_hostcheck.py can compare
// the exact source the HIP kernels execute per lane against the oracle WITHOUT a GPU.
// It is not part of the product and nothing in reduced-3dgs_amd/ links it; the cooperative kernel
// skeletons (LDS staging, DPP reductions, atomics, sorts) are only exercised by the -m gpu tests.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../reduced-3dgs_amd/csrc/blend_math.h"
#include "../../reduced-3dgs_amd/csrc/common.h"   // (includes gauss_math.h)

using namespace r3;

static Camera make_cam(const float* view, const float* proj, const float* campos, int W, int H, float tanx, float tany,
                       float mod)
{
    Camera c;
    for (int k = 0; k < 16; k++) {
        c.view[k] = view[k];
        c.proj[k] = proj[k];
    }
    for (int k = 0; k < 3; k++) c.campos[k] = campos[k];
    c.tan_fovx = tanx;
    c.tan_fovy = tany;
    c.focal_y = H / (2.0f * tany);
    c.focal_x = W / (2.0f * tanx);
    c.W = W;
    c.H = H;
    c.gx = (W + kTile - 1) / kTile;
    c.gy = (H + kTile - 1) / kTile;
    c.scale_modifier = mod;
    return c;
}

extern "C" {

void hc_preprocess(int P, int M, const int* degs, const float* means, const float* scales, float mod, const float* rots,
                   const float* opac, const float* shs, const float* cov_pre, const float* col_pre, const float* view,
                   const float* proj, const float* campos, int W, int H, float tanx, float tany, int* radii, float* xy,
                   float* depths, float* conic_op, float* rgb, unsigned* clamp_bits, unsigned* tiles, int* rect)
{
    const Camera cam = make_cam(view, proj, campos, W, H, tanx, tany, mod);
    for (int i = 0; i < P; i++) {
        PreOut o;
        float sc[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0};
        if (!cov_pre) {
            memcpy(sc, scales + 3 * i, 12);
            memcpy(q, rots + 4 * i, 16);
        }
        preprocess_one(cam, means[3 * i], means[3 * i + 1], means[3 * i + 2], sc, q, cov_pre ? cov_pre + 6 * i : nullptr,
                       opac[i], &o);
        radii[i] = o.radius;
        tiles[i] = o.tiles;
        if (o.radius <= 0) continue;
        xy[2 * i] = o.px;
        xy[2 * i + 1] = o.py;
        depths[i] = o.depth;
        conic_op[4 * i] = o.conic[0];
        conic_op[4 * i + 1] = o.conic[1];
        conic_op[4 * i + 2] = o.conic[2];
        conic_op[4 * i + 3] = o.opacity;
        rect[4 * i] = o.rmin[0];
        rect[4 * i + 1] = o.rmin[1];
        rect[4 * i + 2] = o.rmax[0];
        rect[4 * i + 3] = o.rmax[1];
        if (col_pre) {
            memcpy(rgb + 3 * i, col_pre + 3 * i, 12);
            clamp_bits[i] = 0;
        } else {
            ShRowPlain row{shs + 3 * (size_t)M * i};
            sh_to_rgb(degs[i], row, means[3 * i], means[3 * i + 1], means[3 * i + 2], cam.campos, rgb + 3 * i,
                      clamp_bits + i);
        }
    }
}

// The product's opacity-aware tile rects (gauss_math.h tighten_rect) for a whole scene: rects [P][4] uint16 (x0, y0, x1,
// y1; rows of culled Gaussians zero), tiles [P] = their areas.  tests/test_hostcheck.py checks with the oracle that the
// tiles they leave out of the reference's squares hold no pixel the reference would blend.
void hc_tight_rects(int P, const float* means, const float* scales, float mod, const float* rots, const float* opac,
                    const float* view, const float* proj, const float* campos, int W, int H, float tanx, float tany,
                    int* radii, unsigned short* rects, unsigned* tiles, unsigned* tiles_ref)
{
    const Camera cam = make_cam(view, proj, campos, W, H, tanx, tany, mod);
    for (int i = 0; i < P; i++) {
        PreOut o;
        preprocess_one(cam, means[3 * i], means[3 * i + 1], means[3 * i + 2], scales + 3 * i, rots + 4 * i, nullptr, opac[i],
                       &o, true);
        radii[i] = o.radius;
        tiles[i] = o.tiles;
        tiles_ref[i] = o.tiles_ref;
        for (int k = 0; k < 4; k++) rects[4 * i + k] = 0;
        if (o.radius <= 0) continue;
        rects[4 * i] = (unsigned short)o.rmin[0];
        rects[4 * i + 1] = (unsigned short)o.rmin[1];
        rects[4 * i + 2] = (unsigned short)o.rmax[0];
        rects[4 * i + 3] = (unsigned short)o.rmax[1];
    }
}

static Splat splat_of(const float* xy, const float* conic_op, const float* rgb, unsigned id)
{
    Splat s;
    s.x = xy[2 * id];
    s.y = xy[2 * id + 1];
    s.cA = conic_op[4 * id];
    s.cB = conic_op[4 * id + 1];
    s.cC = conic_op[4 * id + 2];
    s.op = conic_op[4 * id + 3];
    s.r = rgb[3 * id];
    s.g = rgb[3 * id + 1];
    s.b = rgb[3 * id + 2];
    return s;
}

// use_region != 0: apply the kernels' per-(entry, 8x8 quadrant) pre-test exactly as blend.hip does; the
// outputs must be identical with and without it.  *skipped / *total count (entry, quadrant) pairs.
void hc_blend_fwd(int W, int H, const unsigned* ranges, const unsigned* point_list, const float* xy, const float* rgb,
                  const float* conic_op, const float* bg, float* out_color, float* final_T, unsigned* n_contrib,
                  int use_region, long* skipped, long* total)
{
    const int gx = (W + kTile - 1) / kTile;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / kTile) * gx + px / kTile;
            const float qx0 = (float)((px / 8) * 8), qy0 = (float)((py / 8) * 8);
            FwdPix p;
            fwd_pix_init(p, true);
            for (unsigned k = ranges[2 * tile]; k < ranges[2 * tile + 1]; k++) {
                float Tb;
                const Splat s = splat_of(xy, conic_op, rgb, point_list[k]);
                if (use_region) {
                    const bool keep = region_may_contribute(s, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
                    if (px % 8 == 0 && py % 8 == 0) {
                        if (total) ++*total;
                        if (skipped && !keep) ++*skipped;
                    }
                    if (!keep) continue;
                }
                if (fwd_step(s, (float)px, (float)py, k - ranges[2 * tile] + 1, p, &Tb) == 2) break;
            }
            const size_t pix = (size_t)W * py + px, plane = (size_t)W * H;
            const float T = fwd_pix_T(p);
            final_T[pix] = T;
            n_contrib[pix] = p.last;
            out_color[pix] = p.C0 + T * bg[0];
            out_color[plane + pix] = p.C1 + T * bg[1];
            out_color[2 * plane + pix] = p.C2 + T * bg[2];
        }
}

// Statistics of the (list entry, 8x8 quadrant) pairs of a forward pass (design aid): out[0] = all pairs, out[1] = pairs the
// region pre-test keeps, out[2] = kept pairs in which at least one pixel really blends the entry (alpha >= 1/255 and the
// pixel not saturated), out[3] = pairs with a blending pixel (must equal out[2]: the pre-test is conservative),
// out[4] = (entry, tile) pairs with any blending pixel, out[5] = all (entry, tile) pairs, out[6] = kept pairs the forward
// kernel evaluates (until every pixel of the quadrant is saturated), out[7] = kept pairs the backward kernel evaluates
// (in front of the quadrant's deepest contributor).
void hc_pair_stats(int W, int H, const unsigned* ranges, const unsigned* point_list, const float* xy, const float* rgb,
                   const float* conic_op, long* out)
{
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    for (int k = 0; k < 8; k++) out[k] = 0;
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            const unsigned r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            FwdPix pix[256];
            for (int i = 0; i < 256; i++) fwd_pix_init(pix[i], tx * 16 + (i & 15) < W && ty * 16 + (i >> 4) < H);
            long kept_before[4] = {0, 0, 0, 0}, kept_at_last[4] = {0, 0, 0, 0};
            for (unsigned k = r0; k < r1; k++) {
                const Splat s = splat_of(xy, conic_op, rgb, point_list[k]);
                bool tile_hit = false;
                for (int q = 0; q < 4; q++) {
                    const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
                    const bool keep = region_may_contribute(s, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
                    bool hit = false, live = false;
                    for (int j = 0; j < 64; j++) live |= fwd_pix_live(pix[((q >> 1) * 8 + (j >> 3)) * 16 + (q & 1) * 8 + (j & 7)]);
                    if (keep && live) out[6]++;
                    kept_before[q] += keep;
                    for (int j = 0; j < 64; j++) {
                        const int lx = (q & 1) * 8 + (j & 7), ly = (q >> 1) * 8 + (j >> 3);
                        float Tb;
                        if (fwd_step(s, (float)(tx * 16 + lx), (float)(ty * 16 + ly), k - r0 + 1, pix[ly * 16 + lx], &Tb) == 1) hit = true;
                    }
                    out[0]++;
                    out[1] += keep;
                    out[2] += keep && hit;
                    out[3] += hit;
                    tile_hit |= hit;
                    if (hit) kept_at_last[q] = kept_before[q];
                }
                out[4] += tile_hit;
                out[5]++;
            }
            for (int q = 0; q < 4; q++) out[7] += kept_at_last[q];
        }
}

// Lane utilisation of the two blend kernels (design aid, tools/lane_utilisation.py): for every (list entry, 8x8 quadrant)
// pair the kernels EVALUATE -- forward: the region pre-test keeps it and a pixel of the quadrant is still live; backward:
// kept and not behind the quadrant's deepest contributor -- how many of the wave's 64 lanes do useful work there
// (forward: alpha >= 1/255, power <= 0 and the pixel still live; backward: bwd_test valid).
//   fwd_hist[0..64], bwd_hist[0..64]: evaluated pairs by number of useful lanes;
// and what a 4x4-granular pre-test with per-DPP-row entry lists would evaluate instead (lanes remapped so that a 16-lane
// row owns a 4x4 pixel block; each row walks only the entries whose pre-test keeps ITS block; the wave's trip count is
// the longest of its four rows' lists):
//   blk[0] = sum over evaluated (entry, quadrant) of 1 (= today's trips), blk[1] / blk[2] = forward / backward trips with
//   per-row lists (sum over quadrants and 64-entry chunks of max over the 4 rows of its kept entries of the chunk that
//   are evaluated), blk[3] / blk[4] = sum over rows of kept (entry, 4x4 block) pairs evaluated by the forward / backward
//   (the lane-work that remains), blk[5] = backward evaluations today.
void hc_lane_utilisation(int W, int H, const unsigned* ranges, const unsigned* point_list, const float* xy, const float* rgb,
                         const float* conic_op, const unsigned* n_contrib, long* fwd_hist, long* bwd_hist, long* blk)
{
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    for (int k = 0; k <= 64; k++) fwd_hist[k] = bwd_hist[k] = 0;
    for (int k = 0; k < 6; k++) blk[k] = 0;
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            const unsigned r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            FwdPix pix[256];
            unsigned qlast[4] = {0, 0, 0, 0};
            for (int i = 0; i < 256; i++) {
                const int x = tx * 16 + (i & 15), y = ty * 16 + (i >> 4);
                const bool in = x < W && y < H;
                fwd_pix_init(pix[i], in);
                if (in) {
                    const int q = ((i >> 4) / 8) * 2 + (i & 15) / 8;
                    const unsigned nc = n_contrib[(size_t)W * y + x];
                    if (nc > qlast[q]) qlast[q] = nc;
                }
            }
            // per quadrant and 64-entry chunk: kept entries of each 4x4 block that the forward / backward evaluate
            long f_rows[4][4], b_rows[4][4];
            for (unsigned k = r0; k < r1; k++) {
                const unsigned pos = k - r0;
                if ((pos & 63u) == 0u)
                    for (int q = 0; q < 4; q++)
                        for (int b = 0; b < 4; b++) f_rows[q][b] = b_rows[q][b] = 0;
                const Splat s = splat_of(xy, conic_op, rgb, point_list[k]);
                const QSplat qs = scale_splat(s);
                for (int q = 0; q < 4; q++) {
                    const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
                    const bool keep = region_may_contribute(s, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
                    bool live = false;
                    for (int j = 0; j < 64; j++) live |= fwd_pix_live(pix[((q >> 1) * 8 + (j >> 3)) * 16 + (q & 1) * 8 + (j & 7)]);
                    const bool f_eval = keep && live, b_eval = keep && pos < qlast[q];
                    bool keep_b[4];
                    for (int b = 0; b < 4; b++) {
                        const float bx0 = qx0 + (float)((b & 1) * 4), by0 = qy0 + (float)((b >> 1) * 4);
                        keep_b[b] = region_may_contribute(s, bx0, bx0 + 3.f, by0, by0 + 3.f);
                    }
                    int f_lanes = 0, b_lanes = 0;
                    bool live_b[4] = {false, false, false, false};
                    for (int j = 0; j < 64; j++) {
                        const int lx = (q & 1) * 8 + (j & 7), ly = (q >> 1) * 8 + (j >> 3);
                        const int b = ((j >> 3) / 4) * 2 + (j & 7) / 4;
                        FwdPix& px = pix[ly * 16 + lx];
                        live_b[b] |= fwd_pix_live(px);
                        const float fx = (float)(tx * 16 + lx), fy = (float)(ty * 16 + ly);
                        bool inb;
                        const float alpha = fwd_alpha(qs, fx, fy, &inb);
                        const bool vis = inb && alpha >= 1.0f / 255.0f;
                        if (vis && fwd_pix_live(px)) f_lanes++;
                        const bool inside = tx * 16 + lx < W && ty * 16 + ly < H;
                        if (vis && inside && pos < n_contrib[(size_t)W * (ty * 16 + ly) + tx * 16 + lx]) b_lanes++;
                        float Tb;
                        if (keep) fwd_step(qs, fx, fy, pos + 1, px, &Tb);
                    }
                    if (f_eval) {
                        fwd_hist[f_lanes]++;
                        blk[0]++;
                        for (int b = 0; b < 4; b++)
                            if (keep_b[b] && live_b[b]) {
                                f_rows[q][b]++;
                                blk[3]++;
                            }
                    }
                    if (b_eval) {
                        bwd_hist[b_lanes]++;
                        blk[5]++;
                        for (int b = 0; b < 4; b++)
                            if (keep_b[b]) {   // (a per-block deepest contributor would cut a little more)
                                b_rows[q][b]++;
                                blk[4]++;
                            }
                    }
                }
                if ((pos & 63u) == 63u || k + 1 == r1)
                    for (int q = 0; q < 4; q++) {
                        long fm = 0, bm = 0;
                        for (int b = 0; b < 4; b++) {
                            if (f_rows[q][b] > fm) fm = f_rows[q][b];
                            if (b_rows[q][b] > bm) bm = b_rows[q][b];
                        }
                        blk[1] += fm;
                        blk[2] += bm;
                    }
            }
        }
}

// Three alternative lane mappings of the blend kernels, priced on the CPU before any is built (design aid, round 6:
// tools/lane_utilisation.py, profiles/r06_lane_utilisation.txt).  Forward = the entry is kept by the (conservative) region
// pre-test of the region in question and a pixel of that region is still live; useful lane = alpha >= 1/255 on a live pixel.
// Backward = kept and in front of the region's deepest contributor; useful lane = in front of the PIXEL's last contributor
// and alpha >= 1/255.  out[] (longs), f = forward, b = backward (b at index + 20):
//   [0]  today's trips: evaluated (entry, 8x8 quadrant) pairs            [1]  useful lane-evaluations (independent of the mapping)
//  (i) 64-pixel footprints chosen by the splat's orientation -- 8x8 quadrants, 16x4 strips or 4x16 strips, four of each per tile:
//   [2]  trips if every ENTRY could pick its best footprint (the T recurrence forbids it: a wave's pixels would change under
//        it; a lower bound)                                               [3]  trips if every TILE picks one footprint for all its entries
//   [4]  trips with 16x4 strips everywhere                                [5]  trips with 4x16 strips everywhere
//  (ii) two-phase walk: phase A evaluates alpha over compacted (entry, 4x4 block) items, four items per wave instruction;
//       phase B runs the sequential T recurrence per PIXEL over that pixel's own survivors:
//   [6]  phase-A items (kept (entry, 4x4 block) pairs)                    [7]  phase-B trips: per 8x8 quadrant wave the most survivors any of its
//                                                                             pixels has, summed (lanes walk their own lists)
//   [8]  survivors in all (= useful lane-evaluations: the LDS records phase A writes and phase B reads)
//  (iii) one 16-lane DPP row owns a whole tile (16 pixels per lane: slot k = 4x4 block k), a wave walks FOUR tiles' lists side by
//        side (2x2 tile groups), so that an entry's nine sums come from one row -- no cross-row combination:
//   [9]  trips (per group and 64-entry chunk position: the longest of the four lists)
//   [10] slots executed (per trip: the 4x4 blocks ANY of the four rows' entries keeps; a slot costs one evaluation of 64 lanes)
//   [11] (entry, tile) pairs with at least one kept block (= the row-local reductions, one per row and trip instead of one wave
//        reduction per pair)
void hc_lane_variants(int W, int H, const unsigned* ranges, const unsigned* point_list, const float* xy, const float* rgb,
                      const float* conic_op, const unsigned* n_contrib, long* out)
{
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    for (int k = 0; k < 40; k++) out[k] = 0;
    // region r of shape s: s = 0: 8x8 quadrants, 1: 16 wide x 4 tall strips, 2: 4 wide x 16 tall strips
    auto region_of = [](int s, int lx, int ly) { return s == 0 ? (ly / 8) * 2 + lx / 8 : s == 1 ? ly / 4 : lx / 4; };
    auto region_box = [](int s, int r, int* x0, int* x1, int* y0, int* y1) {
        if (s == 0) { *x0 = (r & 1) * 8; *x1 = *x0 + 7; *y0 = (r >> 1) * 8; *y1 = *y0 + 7; }
        else if (s == 1) { *x0 = 0; *x1 = 15; *y0 = r * 4; *y1 = *y0 + 3; }
        else { *x0 = r * 4; *x1 = *x0 + 3; *y0 = 0; *y1 = 15; }
    };
    // per tile of a 2x2 group: for every list position the 16-bit mask of kept 4x4 blocks (forward / backward), for (iii)
    std::vector<unsigned short> fmask[4], bmask[4];
    for (int gy2 = 0; gy2 < (gy + 1) / 2; gy2++)
        for (int gx2 = 0; gx2 < (gx + 1) / 2; gx2++) {
            for (int m = 0; m < 4; m++) { fmask[m].clear(); bmask[m].clear(); }
            for (int m = 0; m < 4; m++) {
                const int tx = gx2 * 2 + (m & 1), ty = gy2 * 2 + (m >> 1);
                if (tx >= gx || ty >= gy) continue;
                const int tile = ty * gx + tx;
                const unsigned r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
                FwdPix pix[256];
                unsigned nc[256], last_of[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, last_blk[16];
                long surv_f[256], surv_b[256];
                for (int b = 0; b < 16; b++) last_blk[b] = 0;
                for (int i = 0; i < 256; i++) {
                    const int lx = i & 15, ly = i >> 4, x = tx * 16 + lx, y = ty * 16 + ly;
                    const bool in = x < W && y < H;
                    fwd_pix_init(pix[i], in);
                    nc[i] = in ? n_contrib[(size_t)W * y + x] : 0u;
                    surv_f[i] = surv_b[i] = 0;
                    for (int s_ = 0; s_ < 3; s_++) {
                        const int r = region_of(s_, lx, ly);
                        if (nc[i] > last_of[s_][r]) last_of[s_][r] = nc[i];
                    }
                    const int b = (ly / 4) * 4 + lx / 4;
                    if (nc[i] > last_blk[b]) last_blk[b] = nc[i];
                }
                long tile_trips_f[3] = {0, 0, 0}, tile_trips_b[3] = {0, 0, 0};
                for (unsigned k = r0; k < r1; k++) {
                    const unsigned pos = k - r0;
                    const Splat s = splat_of(xy, conic_op, rgb, point_list[k]);
                    const QSplat qs = scale_splat(s);
                    // live pixels per region BEFORE this entry, then the per-pixel decisions
                    bool live_r[3][4] = {{false}}, live_b[16];
                    for (int b = 0; b < 16; b++) live_b[b] = false;
                    for (int i = 0; i < 256; i++)
                        if (fwd_pix_live(pix[i])) {
                            const int lx = i & 15, ly = i >> 4;
                            for (int s_ = 0; s_ < 3; s_++) live_r[s_][region_of(s_, lx, ly)] = true;
                            live_b[(ly / 4) * 4 + lx / 4] = true;
                        }
                    bool keep_q0 = false;
                    long nf[3] = {0, 0, 0}, nb[3] = {0, 0, 0};
                    for (int s_ = 0; s_ < 3; s_++)
                        for (int r = 0; r < 4; r++) {
                            int x0, x1, y0, y1;
                            region_box(s_, r, &x0, &x1, &y0, &y1);
                            const bool keep = region_may_contribute(s, (float)(tx * 16 + x0), (float)(tx * 16 + x1),
                                                                    (float)(ty * 16 + y0), (float)(ty * 16 + y1));
                            if (s_ == 0) keep_q0 |= keep;
                            nf[s_] += keep && live_r[s_][r];
                            nb[s_] += keep && pos < last_of[s_][r];
                        }
                    out[0] += nf[0];
                    out[20] += nb[0];
                    out[2] += std::min(nf[0], std::min(nf[1], nf[2]));
                    out[22] += std::min(nb[0], std::min(nb[1], nb[2]));
                    out[4] += nf[1];
                    out[24] += nb[1];
                    out[5] += nf[2];
                    out[25] += nb[2];
                    for (int s_ = 0; s_ < 3; s_++) { tile_trips_f[s_] += nf[s_]; tile_trips_b[s_] += nb[s_]; }
                    unsigned short mf = 0, mb = 0;
                    for (int b = 0; b < 16; b++) {
                        const float bx0 = (float)(tx * 16 + (b & 3) * 4), by0 = (float)(ty * 16 + (b >> 2) * 4);
                        const bool keep = region_may_contribute(s, bx0, bx0 + 3.f, by0, by0 + 3.f);
                        if (keep && live_b[b]) { mf |= (unsigned short)(1u << b); out[6]++; }
                        if (keep && pos < last_blk[b]) { mb |= (unsigned short)(1u << b); out[26]++; }
                    }
                    fmask[m].push_back(mf);
                    bmask[m].push_back(mb);
                    out[11] += mf != 0;
                    out[31] += mb != 0;
                    for (int i = 0; i < 256; i++) {
                        const int lx = i & 15, ly = i >> 4;
                        const float fx = (float)(tx * 16 + lx), fy = (float)(ty * 16 + ly);
                        bool inb;
                        const float alpha = fwd_alpha(qs, fx, fy, &inb);
                        const bool vis = inb && alpha >= 1.0f / 255.0f;
                        if (vis && fwd_pix_live(pix[i])) { out[1]++; surv_f[i]++; }
                        if (vis && pos < nc[i]) { out[21]++; surv_b[i]++; }
                        float Tb;
                        if (keep_q0) fwd_step(qs, fx, fy, pos + 1, pix[i], &Tb);
                    }
                }
                out[3] += std::min(tile_trips_f[0], std::min(tile_trips_f[1], tile_trips_f[2]));
                out[23] += std::min(tile_trips_b[0], std::min(tile_trips_b[1], tile_trips_b[2]));
                for (int q = 0; q < 4; q++) {
                    long mf = 0, mb = 0;
                    for (int j = 0; j < 64; j++) {
                        const int i = ((q >> 1) * 8 + (j >> 3)) * 16 + (q & 1) * 8 + (j & 7);
                        mf = std::max(mf, surv_f[i]);
                        mb = std::max(mb, surv_b[i]);
                        out[8] += surv_f[i];
                        out[28] += surv_b[i];
                    }
                    out[7] += mf;
                    out[27] += mb;
                }
            }
            // (iii): each row walks the entries of ITS tile that keep at least one block, the four rows side by side
            for (int dir = 0; dir < 2; dir++) {
                std::vector<unsigned short> rows[4];
                size_t longest = 0;
                for (int m = 0; m < 4; m++) {
                    for (unsigned short v : (dir ? bmask[m] : fmask[m]))
                        if (v) rows[m].push_back(v);
                    longest = std::max(longest, rows[m].size());
                }
                out[9 + 20 * dir] += (long)longest;
                for (size_t t = 0; t < longest; t++) {
                    unsigned any = 0;
                    for (int m = 0; m < 4; m++)
                        if (t < rows[m].size()) any |= rows[m][t];
                    out[10 + 20 * dir] += __builtin_popcount(any);
                }
            }
        }
}

// How many list entries a TILE-level region test (the blend kernels' region_may_contribute over the whole 16x16 tile) would
// drop before they are binned (design aid, round 6): out[0] = list entries, out[1] = entries whose tile test says "cannot
// contribute", out[2] = of those, entries of Gaussians whose rect has at most 32 tiles (a 32-bit mask could carry the decision).
void hc_tile_test_stats(int W, int H, const unsigned* ranges, const unsigned* point_list, const float* xy, const float* rgb,
                        const float* conic_op, const unsigned* tiles_of_gaussian, long* out)
{
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    out[0] = out[1] = out[2] = 0;
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            for (unsigned k = ranges[2 * tile]; k < ranges[2 * tile + 1]; k++) {
                const unsigned id = point_list[k];
                const Splat s = splat_of(xy, conic_op, rgb, id);
                const bool keep = region_may_contribute(s, (float)(tx * 16), (float)(tx * 16 + 15), (float)(ty * 16), (float)(ty * 16 + 15));
                out[0]++;
                if (!keep) {
                    out[1]++;
                    if (tiles_of_gaussian[id] <= 32u) out[2]++;
                }
            }
        }
}

// acc: double[P][9] = mx, my, cA, cB, cC, op, r, g, b  (mx,my already scaled by 0.5W / 0.5H)
void hc_blend_bwd(int P, int W, int H, const unsigned* ranges, const unsigned* point_list, const float* bg,
                  const float* xy, const float* conic_op, const float* rgb, const float* final_T,
                  const unsigned* n_contrib, const float* dL_dpix, double* acc, int use_region)
{
    (void)P;
    const int gx = (W + kTile - 1) / kTile;
    const size_t plane = (size_t)W * H;
    const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / kTile) * gx + px / kTile;
            const size_t pix = (size_t)W * py + px;
            BwdPix p;
            const float g0 = dL_dpix[pix], g1 = dL_dpix[plane + pix], g2 = dL_dpix[2 * plane + pix];
            bwd_pix_init(p, final_T[pix], n_contrib[pix], g0, g1, g2, bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
            for (long pos = (long)p.last - 1; pos >= 0; pos--) {
                const unsigned id = point_list[ranges[2 * tile] + pos];
                const Splat s = splat_of(xy, conic_op, rgb, id);
                if (use_region) {
                    const float qx0 = (float)((px / 8) * 8), qy0 = (float)((py / 8) * 8);
                    if (!region_may_contribute(s, qx0, qx0 + 7.f, qy0, qy0 + 7.f)) continue;
                }
                SplatGrad g;
                g.mx = g.my = g.cA = g.cB = g.cC = g.op = g.r = g.g = g.b = 0.f;
                if (bwd_step(s, (float)px, (float)py, (unsigned)pos, p, g)) {
                    double* a = acc + 9 * (size_t)id;
                    a[0] += (double)(g.mx * half_w);
                    a[1] += (double)(g.my * half_h);
                    a[2] += g.cA;
                    a[3] += g.cB;
                    a[4] += g.cC;
                    a[5] += g.op;
                    a[6] += g.r;
                    a[7] += g.g;
                    a[8] += g.b;
                }
            }
        }
}

// The backward blend walked in LIST SEGMENTS of S entries from the forward's checkpoints, pixel by pixel, with the kernels'
// own per-lane functions and the kernels' own arithmetic for the state at a segment's end (blend.hip: blend_fwd_kernel parks
// (T, C0, C1, C2) in front of entry k S of a tile's list and the colour it ends with; a unit of blend_bwd_kernel that ends in
// front of a pixel's last contributor starts it from the checkpoint at its end).  A tile deeper than S is walked in
// n = min(ceil(deepest contributor / S), max_seg) segments, the last one taking the rest; the segments run in ascending order
// here (any order gives the same sums: every entry belongs to one segment).  acc as hc_blend_bwd; *n_split = pixels that
// started at least one segment from a checkpoint.
void hc_blend_bwd_segments(int P, int W, int H, const unsigned* ranges, const unsigned* point_list, const float* bg,
                           const float* xy, const float* conic_op, const float* rgb, const float* final_T,
                           const unsigned* n_contrib, const float* dL_dpix, double* acc, int S, int max_seg, long* n_split)
{
    (void)P;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const size_t plane = (size_t)W * H;
    const float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
    *n_split = 0;
    std::vector<float> ck;   // checkpoints of one pixel: 4 floats per k >= 1
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            unsigned deepest = 0;
            for (int py = ty * kTile; py < (ty + 1) * kTile && py < H; py++)
                for (int px = tx * kTile; px < (tx + 1) * kTile && px < W; px++)
                    deepest = n_contrib[(size_t)W * py + px] > deepest ? n_contrib[(size_t)W * py + px] : deepest;
            unsigned nseg = deepest > (unsigned)S ? (deepest + (unsigned)S - 1u) / (unsigned)S : 1u;
            if (nseg > (unsigned)max_seg) nseg = (unsigned)max_seg;
            for (int py = ty * kTile; py < (ty + 1) * kTile && py < H; py++)
                for (int px = tx * kTile; px < (tx + 1) * kTile && px < W; px++) {
                    const size_t pix = (size_t)W * py + px;
                    const unsigned last = n_contrib[pix];
                    // the forward of this pixel, leaving its checkpoints
                    FwdPix f;
                    fwd_pix_init(f, true);
                    ck.assign(4 * ((size_t)last / (size_t)S + 2), 0.f);
                    for (unsigned pos = 0; pos < last; pos++) {
                        if (pos != 0 && pos % (unsigned)S == 0) {
                            float* c = &ck[4 * (pos / (unsigned)S)];
                            c[0] = f.T; c[1] = f.C0; c[2] = f.C1; c[3] = f.C2;
                        }
                        const Splat s = splat_of(xy, conic_op, rgb, point_list[ranges[2 * tile] + pos]);
                        float tb;
                        fwd_step(s, (float)px, (float)py, pos + 1u, f, &tb);
                    }
                    const float g0 = dL_dpix[pix], g1 = dL_dpix[plane + pix], g2 = dL_dpix[2 * plane + pix];
                    bool from_ckpt = false;
                    for (unsigned k = 0; k < nseg; k++) {
                        const unsigned lo = k * (unsigned)S;
                        const unsigned hi = k + 1u < nseg ? lo + (unsigned)S : 0xFFFFFFFFu;
                        if (lo >= last) break;
                        BwdPix p;
                        bwd_pix_init(p, final_T[pix], last, g0, g1, g2, bg[0] * g0 + bg[1] * g1 + bg[2] * g2);
                        unsigned top = last;
                        if (hi < last) {   // passes through this segment's end: blend_bwd_kernel's checkpoint start
                            const float* c = &ck[4 * (hi / (unsigned)S)];
                            p.A = ((f.C0 - c[1]) * p.g0 + (f.C1 - c[2]) * p.g1 + (f.C2 - c[3]) * p.g2 + p.T * p.A) * R3_RCP(c[0]);
                            p.T = c[0];
                            top = hi;
                            from_ckpt = true;
                        }
                        for (long pos = (long)top - 1; pos >= (long)lo; pos--) {
                            const unsigned id = point_list[ranges[2 * tile] + pos];
                            const Splat s = splat_of(xy, conic_op, rgb, id);
                            SplatGrad g;
                            g.mx = g.my = g.cA = g.cB = g.cC = g.op = g.r = g.g = g.b = 0.f;
                            if (bwd_step(s, (float)px, (float)py, (unsigned)pos, p, g)) {
                                double* a = acc + 9 * (size_t)id;
                                a[0] += (double)(g.mx * half_w);
                                a[1] += (double)(g.my * half_h);
                                a[2] += g.cA;
                                a[3] += g.cB;
                                a[4] += g.cC;
                                a[5] += g.op;
                                a[6] += g.r;
                                a[7] += g.g;
                                a[8] += g.b;
                            }
                        }
                    }
                    *n_split += from_ckpt ? 1 : 0;
                }
        }
}

// Fuzz the conservativeness of region_may_contribute(): random (also extremely anisotropic) splats against
// random 8x8 pixel blocks; a violation = the pre-test says "skip" although some pixel's fwd_step would blend.
// Returns the number of violations; *n_skip counts skipped blocks, *n_tight blocks kept with no contributor.
long hc_region_fuzz(long n, unsigned seed, long* n_skip, long* n_keep_empty)
{
    unsigned long long st = seed * 2654435761ull + 88172645463325252ull;
    auto rnd = [&]() {  // xorshift64*, uniform in [0,1)
        st ^= st >> 12;
        st ^= st << 25;
        st ^= st >> 27;
        return (float)(((st * 2685821657736338717ull) >> 40) & 0xFFFFFF) / 16777216.0f;
    };
    long bad = 0;
    for (long it = 0; it < n; it++) {
        // random covariance: eigenvalues spanning 1e-1 .. 1e5 px^2 (+0.3 low-pass like the rasterizer), any angle
        const float l1 = expf(rnd() * 13.8f - 2.3f), l2 = l1 * expf(-rnd() * 11.5f);
        const float th = rnd() * 6.2831853f, c = cosf(th), s_ = sinf(th);
        const float a = c * c * l1 + s_ * s_ * l2 + 0.3f, b = c * s_ * (l1 - l2), d = s_ * s_ * l1 + c * c * l2 + 0.3f;
        const float det = a * d - b * b;
        Splat s;
        s.cA = d / det;
        s.cB = -b / det;
        s.cC = a / det;
        s.op = rnd() < 0.1f ? rnd() * 0.01f : rnd();
        const float reach = 4.f * sqrtf(l1) + 12.f;
        s.x = 100.f + (rnd() * 2.f - 1.f) * reach;
        s.y = 100.f + (rnd() * 2.f - 1.f) * reach;
        s.r = s.g = s.b = 0.5f;
        const float X0 = 96.f, Y0 = 96.f;
        const bool keep = region_may_contribute(s, X0, X0 + 7.f, Y0, Y0 + 7.f);
        bool any = false;
        for (int py = 0; py < 8 && !any; py++)
            for (int px = 0; px < 8; px++) {
                FwdPix p;
                fwd_pix_init(p, true);
                float Tb;
                if (fwd_step(s, X0 + px, Y0 + py, 1, p, &Tb) != 0) {
                    any = true;
                    break;
                }
            }
        if (!keep && any) bad++;
        if (!keep) ++*n_skip;
        if (keep && !any) ++*n_keep_empty;
    }
    return bad;
}

void hc_truncated_colours(int P, int M, int nslots, const int* degs, const float* means, const float* campos,
                          const float* shs, float* colours /* [P][nslots][3], pre-zeroed */)
{
    for (int i = 0; i < P; i++) {
        ShRowPlain row{shs + 3 * (size_t)M * i};
        sh_truncated_colours(degs[i], nslots, row, means[3 * i], means[3 * i + 1], means[3 * i + 2], campos,
                             colours + 3 * (size_t)nslots * i);
    }
}

static long g_cached_mismatches = 0;
long hc_cached_sh_mismatches(void) { return g_cached_mismatches; }

static int g_f64_chain = 0;
// 1: the covariance chain in double (the product's default, gauss_math.h cov2d_backward_f64 / cov3d_backward_f64)
void hc_set_f64_chain(int on) { g_f64_chain = on; }

void hc_preprocess_bwd(int P, int M, const int* degs, const float* means, const int* radii, const float* shs,
                       const unsigned* clamp_bits, const float* scales, const float* rots, float mod,
                       const float* cov_pre, const float* view, const float* proj, const float* campos, int W, int H,
                       float tanx, float tany, const float* dL_dmean2D, const float* conic_op, const float* dL_dconic,
                       const float* dL_dcolor, float lambda_sh, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, float* dL_dopacity)
{
    const Camera cam = make_cam(view, proj, campos, W, H, tanx, tany, mod);
    int V = 0;
    for (int i = 0; i < P; i++) V += radii[i] > 0;
    const float mult = lambda_sh != 0.f ? lambda_sh / (float)(V * 15 * 3) : 0.f;
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
        float sc[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, c6[6];
        if (cov_pre)
            memcpy(c6, cov_pre + 6 * i, 24);
        else {
            memcpy(sc, scales + 3 * i, 12);
            memcpy(q, rots + 4 * i, 16);
            cov3d_from_scale_rot(sc, mod, q, c6);
        }
        float dmean[3];
        double dcov6d[6];
        if (g_f64_chain) {
            cov2d_backward_f64(cam, mx, my, mz, c6, dL_dconic[4 * i], dL_dconic[4 * i + 1], dL_dconic[4 * i + 3], dcov6d, dmean);
            for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = (float)dcov6d[k];
        } else {
            cov2d_backward(cam, mx, my, mz, c6, dL_dconic[4 * i], dL_dconic[4 * i + 1], dL_dconic[4 * i + 3],
                           dL_dcov3D + 6 * i, dmean);
        }
        project_backward(cam, mx, my, mz, dL_dmean2D[3 * i], dL_dmean2D[3 * i + 1], dmean);
        if (shs) {
            ShRowPlain row{shs + 3 * (size_t)M * i};
            ShGradPlain sink{dL_dsh + 3 * (size_t)M * i};
            if (mult == 0.f) {
                // as the product does without a sparsity term: the direction derivatives come from the forward
                // (sh_dir_derivs_at); must equal the direct form bit for bit
                float d9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (degs[i] > 0) sh_dir_derivs_at(degs[i], row, mx, my, mz, cam.campos, d9);
                float direct_dsh[48], direct_dmean[3] = {dmean[0], dmean[1], dmean[2]};
                ShGradPlain direct_sink{direct_dsh};
                sh_backward<false>(degs[i], row, direct_sink, nullptr, mx, my, mz, cam.campos, clamp_bits[i], dL_dcolor + 3 * i,
                                   mult, direct_dmean);
                sh_backward<true>(degs[i], row, sink, d9, mx, my, mz, cam.campos, clamp_bits[i], dL_dcolor + 3 * i, mult, dmean);
                const int K = (degs[i] + 1) * (degs[i] + 1);
                g_cached_mismatches += memcmp(direct_dmean, dmean, 12) != 0;
                g_cached_mismatches += memcmp(direct_dsh, dL_dsh + 3 * (size_t)M * i, 12 * (size_t)K) != 0;
            } else {
                sh_backward<false>(degs[i], row, sink, nullptr, mx, my, mz, cam.campos, clamp_bits[i], dL_dcolor + 3 * i, mult,
                                   dmean);
            }
        }
        memcpy(dL_dmean3D + 3 * i, dmean, 12);
        if (!cov_pre) {
            if (g_f64_chain)
                cov3d_backward_f64(sc, mod, q, dcov6d, dL_dscale + 3 * i, dL_drot + 4 * i);
            else
                cov3d_backward(sc, mod, q, dL_dcov3D + 6 * i, dL_dscale + 3 * i, dL_drot + 4 * i);
        }
        dL_dopacity[i] = opacity_backward(dL_dopacity[i], conic_op[4 * i + 3]);
    }
}

// The backward blend's unit lists (common.h TileGrid): tiles[j] = tile of slot j of list `list` (-1: a slot of an edge block
// without a tile); info = {slots, list_tiles_max, bwd_list_fit(pairs), bwd_units_cap(reserve)}.  Returns the slot count.
int hc_unit_list(int gx, int gy, int list, int* tiles, int tiles_cap, unsigned pairs, unsigned reserve, int* info)
{
    const TileGrid g{(uint32_t)gx, (uint32_t)gy};
    const uint32_t n = g.list_slots((uint32_t)list);
    for (uint32_t j = 0; j < n && (int)j < tiles_cap; j++) tiles[j] = (int)g.list_tile((uint32_t)list, j);
    info[0] = (int)n;
    info[1] = (int)g.list_tiles_max();
    info[2] = (int)bwd_list_fit(pairs, g);
    info[3] = (int)bwd_units_cap(reserve, g);
    return (int)n;
}

}  // extern "C"
