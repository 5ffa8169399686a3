"""Timing of the operators either side of the rasterizer at scene scale (one MI355X), HIP events on torch's stream.
Not part of bench.py's metric; prints one JSON line per operator with the algorithmic bytes it must move.

    python tools/reduction_bench.py [--P 500000] [--K 30] [--cameras 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))

import synth_scene as ss  # noqa: E402
from diff_gaussian_rasterization import _C  # noqa: E402
from simple_knn._C import distCUDA2, distIndex2  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        out = fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=500_000)
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--cameras", type=int, default=200)
    a = ap.parse_args()
    P, K, C = a.P, a.K, a.cameras
    cam = ss.make_camera(1600, 1062, 1400.0)
    g = ss.make_gaussians(P, cam, seed=0)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    xyz, scales, rots = dev(g["means3D"]), dev(g["scales"]), dev(g["rotations"])

    def line(name, ms, nbytes, **kw):
        print(json.dumps({"op": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 2),
                          "GBps": round(nbytes / ms / 1e6, 1), **kw}), flush=True)

    ms, mean3 = timed(lambda: distCUDA2(xyz))
    line("distCUDA2", ms, P * 16, P=P)
    ms, (d, idx) = timed(lambda: distIndex2(xyz, K))
    line("distIndex2", ms, P * (12 + 8 * K), P=P, K=K)
    idx = idx.view(P, K)

    cams = [ss.make_camera(1600, 1062, 1400.0, seed=i) for i in range(C)]
    w2ndc = dev(np.stack([c.full_proj_transform for c in cams]))
    w2ndc_inv = torch.linalg.inv(w2ndc)
    Hs = torch.full((C,), 1062, dtype=torch.int32, device="cuda")
    Ws = torch.full((C,), 1600, dtype=torch.int32, device="cuda")
    ms, cube = timed(lambda: _C.find_minimum_projected_pixel_size(w2ndc, w2ndc_inv, xyz, Hs, Ws))
    line("find_minimum_projected_pixel_size", ms, P * 16 + C * 136, P=P, cameras=C)

    radius = cube * (3 ** 0.5) / 2
    ms, (red, mask) = timed(lambda: _C.sphere_ellipsoid_intersection(xyz, scales, rots, idx, radius, K))
    line("sphere_ellipsoid_intersection", ms, P * K * (4 + 1 + 24) + P * 36, P=P, K=K,
         hit_fraction=round(float(mask.float().mean()), 4))
    ms, _ = timed(lambda: _C.allocate_minimum_redundancy_value(red, idx, mask, K))
    line("allocate_minimum_redundancy_value", ms, P * K * 5 + P * 8, P=P, K=K)

    # codebook of one 500k x 45 feature tensor (scene/gaussian_model.py:36-44), 256 centres, tol 1e-4, <= 500 updates
    rng = np.random.default_rng(0)
    vals = dev(rng.normal(0, 0.2, P * 45).astype(np.float32)).view(-1, 1)
    centers = vals[torch.randint(vals.shape[0], (256,), device="cuda")].view(-1)
    t0 = time.perf_counter()
    ids, cen, iters = _C.kmeans_cuda(vals, centers, 1e-4, 500, _want_iterations=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    it = int(iters)
    line("kmeans_cuda", wall, (it + 1) * vals.numel() * 4 + vals.numel() * 4, n_values=vals.numel(), centers=256,
         updates_run=it, ms_per_update=round(wall / max(it, 1), 4))

    # CPU yardstick for the neighbour search: scipy's k-d tree on all host cores, bounded sample of the queries
    from scipy.spatial import cKDTree
    pts = g["means3D"].astype(np.float64)
    t0 = time.perf_counter()
    tree = cKDTree(pts)
    q = pts[: min(P, 100_000)]
    tree.query(q, k=K + 1, workers=-1)
    cpu = time.perf_counter() - t0
    print(json.dumps({"op": "cpu_kdtree_yardstick", "sample_queries": q.shape[0], "seconds": round(cpu, 3),
                      "cores": os.cpu_count()}), flush=True)


if __name__ == "__main__":
    main()
