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
    sys.stdout.flush()


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["metric_500k_1600x1062", "clustered_500k_1600x1062"]):
        run(n)
