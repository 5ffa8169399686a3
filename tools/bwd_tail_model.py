"""Would splitting the heaviest tiles of the backward blend ALONG THEIR LISTS shorten the kernel?  

A processor-sharing model of blend_bwd_kernel's schedule, fed the real per-tile weights of the benchmark shape:
  * a tile's work = the sum over its four 8x8 quadrants of the deepest contributor (ImageState::quad_depth -- the weight
    unit_order_kernel sorts by; computed here with the CPU oracle: max of n_contrib per quadrant);
  * 1024 SIMDs, `slots` single-wave workgroups resident per SIMD (6 at 80 VGPRs), dispatched in launch order into free slots;
  * the waves of a SIMD share its issue capacity equally, but ONE wave cannot use more than `rmax` of it (a wave issues a
    dependent instruction every ~5 cycles against a pipe that takes one every 2.4-4: profiles/r03_valu_rate.txt).
It reproduces what was measured -- row-major launch order 1.25-1.3x slower than heaviest-first (measured 340 -> 258 us) -- and
then answers the question: a split unit costs `overhead` (pixel state + checkpoint loads, a partial first chunk) on top of
its share of the walk.

    python tools/bwd_tail_model.py > profiles/r04_bwd_tail_model.txt
"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth_scene as ss  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def quad_depths(workload="metric_500k_1600x1062"):
    w, cam, g = ss.make_workload(workload)
    W, H = w["W"], w["H"]
    ref = orc.forward(np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = np.zeros((gy * 16, gx * 16), np.int64)
    nc[:H, :W] = ref["state"]["n_contrib"].reshape(H, W)
    rng = ref["state"]["ranges"].astype(np.int64)
    return nc.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5)).transpose(0, 2, 1, 3).reshape(gx * gy, 4), rng[:, 1] - rng[:, 0]


def simulate(units, nsimd=1024, slots=6, rmax=0.45):
    units = list(units)
    act = [[] for _ in range(nsimd)]
    nxt = 0
    for s in range(nsimd * slots):
        if nxt >= len(units):
            break
        act[s % nsimd].append(units[nxt])
        nxt += 1

    def rate(n):
        return min(rmax, 1.0 / n)
    last = [0.0] * nsimd
    ev = [(min(a) / rate(len(a)), s) for s, a in enumerate(act) if a]
    heapq.heapify(ev)
    t = 0.0
    while ev:
        te, s = heapq.heappop(ev)
        if not act[s]:
            continue
        r = rate(len(act[s]))
        tn = last[s] + min(act[s]) / r
        if abs(tn - te) > 1e-9:       # stale event
            heapq.heappush(ev, (tn, s))
            continue
        dt = te - last[s]
        act[s] = [a - dt * r for a in act[s]]
        act[s] = [a for a in act[s] if a > 1e-9]
        last[s] = te
        while nxt < len(units) and len(act[s]) < slots:
            act[s].append(units[nxt])
            nxt += 1
        if act[s]:
            heapq.heappush(ev, (te + min(act[s]) / rate(len(act[s])), s))
        t = max(t, te)
    return t


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "metric_500k_1600x1062"
    q, list_len = quad_depths(workload)
    w = q.sum(1).astype(float)
    lmax = q.max(1)
    hist, edges = np.histogram(w / max(w.mean(), 1e-9), bins=[0, 0.05, 0.25, 0.5, 1, 1.5, 2, 3, 4, 6, 100])
    print(f"{workload}: tile-weight histogram (weight / mean weight): " +
          ", ".join(f"[{a:g},{b:g}) {100 * h / len(w):.1f} %" for a, b, h in zip(edges[:-1], edges[1:], hist)))
    print(f"  work in the heaviest 1 % / 5 % / 10 % of the tiles: " +
          " / ".join(f"{100 * np.sort(w)[::-1][:max(1, int(len(w) * f))].sum() / w.sum():.1f} %" for f in (0.01, 0.05, 0.1)) +
          f"; heaviest tile = {w.max() / (w.sum() / 1024):.3f} of (total work / 1024 SIMDs)")
    print(f"{workload}: {len(w)} tiles; list length mean {list_len.mean():.0f} max {list_len.max()}; deepest contributor per "
          f"tile mean {lmax.mean():.0f} median {np.median(lmax):.0f} p99 {np.percentile(lmax, 99):.0f} max {lmax.max()} -- the "
          f"heaviest tile is {w.max() / w.mean():.2f}x the mean")
    ideal = w.sum() / 1024
    order = np.argsort(-w)
    print("kernel length / (total work / 1024 SIMDs):")
    for rmax in (0.3, 0.45, 0.6):
        row = simulate(w, rmax=rmax) / ideal
        heavy = simulate(w[order], rmax=rmax) / ideal
        line = f"  one wave may use {rmax:.2f} of a SIMD: row-major {row:.3f}  heaviest-first {heavy:.3f} (ratio {row / heavy:.2f}; measured at the metric shape 340 / 258 us = 1.32)"
        for oh in (0.0, 0.05, 0.1):
            parts = []
            for frac in (0.02, 0.1, 0.25, 0.5, 1.0):
                n = int(len(w) * frac)   # the second unit of a split tile pays one more start
                units = np.concatenate([w[order][:n] / 2, w[order][:n] / 2 + oh * w.mean(), w[order][n:]])
                parts.append(f"top {int(frac * 100)} % in two: {simulate(np.sort(units)[::-1], rmax=rmax) / ideal:.3f}")
            line += f"\n      split along the list, extra start = {oh:.2f} x a mean tile's walk: " + ", ".join(parts)
        print(line)
    print("  => see DESIGN.md section 11 for the reading.")


if __name__ == "__main__":
    main()
