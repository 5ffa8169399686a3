"""Lane utilisation of the blend kernels on a named workload, computed on the CPU with the kernels' own per-lane functions
(tests/hostcheck/hostcheck.hip hc_lane_utilisation over blend_math.h; lists = the oracle's over the product's
opacity-aware rects).  No GPU needed: the decisions are per-pixel arithmetic, identical on both sides up to the rounding of
one exp.

    python tools/lane_utilisation.py [workload ...]   ->  stdout (committed as profiles/r05_lane_utilisation.txt)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth_scene as ss  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_hostcheck import _lib, p  # noqa: E402


def tight_rects(L, g, cam, W, H):
    P = g["means3D"].shape[0]
    radii = np.zeros(P, np.int32)
    rects = np.zeros((P, 4), np.uint16)
    tiles = np.zeros(P, np.uint32)
    tiles_ref = np.zeros(P, np.uint32)
    view, proj, campos = (np.ascontiguousarray(a, np.float32) for a in
                          (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
    opac = np.ascontiguousarray(g["opacity"].reshape(-1))
    L.hc_tight_rects(C.c_int(P), p(g["means3D"]), p(g["scales"]), C.c_float(1.0), p(g["rotations"]), p(opac), p(view),
                     p(proj), p(campos), C.c_int(W), C.c_int(H), C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(radii),
                     p(rects), p(tiles), p(tiles_ref))
    return rects


def run(name):
    L = _lib()
    w, cam, g = ss.make_workload(name)
    W, H = w["W"], w["H"]
    bg = np.zeros(3, np.float32)
    rects = tight_rects(L, g, cam, W, H)
    ref = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center, rects=rects)
    st = ref["state"]
    fh, bh, blk = np.zeros(65, np.int64), np.zeros(65, np.int64), np.zeros(6, np.int64)
    L.hc_lane_utilisation(C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["rgb"]),
                          p(st["conic_op"]), p(st["n_contrib"]), p(fh), p(bh), p(blk))
    lanes = np.arange(65)
    print(f"== {name}: {w['P']} Gaussians, {W}x{H}, list entries (opacity-aware rects) {st['pairs']}")
    for tag, h in (("forward  (fwd_alpha passes and the pixel is live)", fh), ("backward (bwd_test valid)", bh)):
        n = h.sum()
        mean = (h * lanes).sum() / max(n, 1)
        cum = np.cumsum(h) / max(n, 1)
        print(f"  {tag}: {n} evaluated (entry, quadrant) pairs, mean useful lanes {mean:.1f} / 64 = {100 * mean / 64:.1f} %")
        print("    useful lanes:   0     1-8    9-16   17-32   33-48   49-63    64")
        edges = [(0, 0), (1, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64)]
        print("    share of pairs: " + "  ".join(f"{100 * h[a:b + 1].sum() / max(n, 1):5.1f}%" for a, b in edges))
        print(f"    median {int(np.searchsorted(cum, 0.5))}, p25 {int(np.searchsorted(cum, 0.25))}, p75 {int(np.searchsorted(cum, 0.75))}")
    print(f"  4x4-granular pre-test, one entry list per 16-lane DPP row (trips = longest row list per quadrant and chunk):")
    print(f"    forward : trips {blk[0]} -> {blk[1]} ({100 * blk[1] / max(blk[0], 1):.1f} %), kept (entry, 4x4 block) pairs "
          f"{blk[3]} = {100 * blk[3] / max(4 * blk[0], 1):.1f} % of today's 4 blocks per trip")
    print(f"    backward: trips {blk[5]} -> {blk[2]} ({100 * blk[2] / max(blk[5], 1):.1f} %), kept (entry, 4x4 block) pairs "
          f"{blk[4]} = {100 * blk[4] / max(4 * blk[5], 1):.1f} %")
    variants(L, name, W, H, st)
    sys.stdout.flush()


# VALU instructions per trip of today's kernels at the metric shape (profiles/r05_pmc_summary.json: 92.3 M / 150.8 M VALU over
# 3.45 M / 3.40 M evaluated (entry, quadrant) pairs) and what the pieces of a trip are taken to cost in the variants.  ASSUMPTIONS,
# stated so that the prices can be redone: alpha evaluation (offsets, falloff, exp, min, two compares) 14; the forward's
# recurrence (test, three FMAs, T update, bookkeeping) 12; the backward's accumulation 21 and its wave reduction + slab
# bookkeeping 23 per reduced entry; an LDS record append / fetch in (ii) 5 each; a vector bit scan + LDS address per row and
# trip in (iii) 8.
I_F, I_B = 26.7, 44.3
C_ALPHA, C_REC_F, C_ACC_B, C_RED_B, C_LIST, C_ROWSCAN = 14.0, 12.0, 21.0, 23.0, 5.0, 8.0


def variants(L, name, W, H, st):
    out = np.zeros(40, np.int64)
    L.hc_lane_variants(C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(st["rgb"]),
                       p(st["conic_op"]), p(st["n_contrib"]), p(out))
    f, b = out[:20], out[20:]
    print("  -- three alternative lane mappings, priced (VERDICT r5 item 5; raw counts from hc_lane_variants, prices from the "
          "stated per-piece costs) --")
    for tag, v, per_trip in (("forward ", f, I_F), ("backward", b, I_B)):
        t0 = max(int(v[0]), 1)
        print(f"  {tag}: today {t0} trips x {per_trip} VALU, {v[1]} useful lane-evaluations = {100 * v[1] / (64 * t0):.1f} % of the lanes")
        print(f"    (i) footprints by orientation: 16x4 strips everywhere {v[4]} trips ({100 * v[4] / t0:.1f} %), 4x16 {v[5]} "
              f"({100 * v[5] / t0:.1f} %); best footprint per TILE {v[3]} ({100 * v[3] / t0:.1f} %); per ENTRY (not realisable: a "
              f"wave's pixels would change under its T recurrence) {v[2]} ({100 * v[2] / t0:.1f} %)")
    # (ii): forward only -- the backward's nine sums are per ENTRY over pixels, lanes that walk their own lists are at
    # different entries
    t0 = max(int(f[0]), 1)
    a_trips = f[6] / 4.0
    cost_ii = a_trips * (C_ALPHA + C_LIST) + f[7] * (C_REC_F + C_LIST)
    print(f"    (ii) two-phase walk, forward: phase A {f[6]} (entry, 4x4 block) items = {a_trips:.0f} trips of four, phase B "
          f"{f[7]} trips (per quadrant: its busiest pixel's survivors; {f[8]} survivor records through LDS, "
          f"{8 * f[8] / max((W * H), 1):.0f} B per pixel): "
          f"{cost_ii / 1e6:.1f} M VALU against {t0 * I_F / 1e6:.1f} M today = {100 * cost_ii / (t0 * I_F):.1f} %  "
          f"(backward: not applicable -- an entry's sums need its pixels at the same entry)")
    for tag, v, per_slot, red in (("forward ", f, I_F, 0.0), ("backward", b, C_ALPHA + C_ACC_B, C_RED_B)):
        t0 = max(int(v[0]), 1)
        today = t0 * (I_F if red == 0.0 else I_B)
        cost = v[10] * per_slot + v[9] * (C_ROWSCAN + red)
        print(f"    (iii) a DPP row owns a tile, four tiles per wave, {tag}: {v[9]} trips, {v[10]} slots of 64 lanes "
              f"({100 * v[1] / max(64 * v[10], 1):.1f} % of the lanes useful), {v[11]} (entry, tile) pairs -> "
              f"{cost / 1e6:.1f} M VALU against {today / 1e6:.1f} M today = {100 * cost / today:.1f} %")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["metric_500k_1600x1062", "clustered_500k_1600x1062"]):
        run(n)
