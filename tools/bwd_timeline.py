"""Debug aid: when and where every workgroup of the backward blend ran (build: python tools/build_variant.py tl blend.hip --
-DR3_TIMELINE; run with R3DGS_LIB=tl) -- or, with the argument `fwd`, of the forward blend (variant tlf, -DR3_TIMELINE_FWD;
one workgroup per 8x8 quadrant there).  Prints the distribution of workgroup durations, the resident workgroups over
time and the per-SIMD load; saves the raw table to gpurun_out/bwd_timeline.npy."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reduced-3dgs_amd")]
FWD = len(sys.argv) > 1 and sys.argv[1] == "fwd"
os.environ.setdefault("R3DGS_LIB", "tlf" if FWD else "tl")
os.environ["R3DGS_GRAPH"] = "0"
import synth_scene as ss  # noqa: E402
from diff_gaussian_rasterization import _C  # noqa: E402

w, cam, g = ss.make_workload(os.environ.get("R3_TL_WORKLOAD", "metric_500k_1600x1062"))
W, H, P = w["W"], w["H"], w["P"]
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = (dv(np.zeros(3, np.float32)), dv(g["means3D"]), torch.Tensor([]), dv(g["opacity"]), dv(g["scales"]), dv(g["rotations"]),
        1.0, torch.Tensor([]), dv(cam.world_view_transform), dv(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, H, W,
        dv(g["sh"]), dv(g["degrees"]), dv(cam.camera_center), False, False)
dl = dv(ss.upstream_grad(W, H, seed=1))
for rep in range(6):   # warm clocks, then keep the last pass's table
    out = _C.rasterize_gaussians(*args)
    R, color, radii, geom, binning, img = out
    _C.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], 1.0, args[7], args[8], args[9], args[10],
                                    args[11], dl, args[14], args[15], args[16], geom, R, binning, img, 0.0, False)
torch.cuda.synchronize()
# forward: one workgroup per 8x8 quadrant; backward: one per (tile, list segment) unit, launched for the most a pass may have
n = ((W + 15) // 16) * ((H + 15) // 16) * 4 if FWD else min(65536, int(_C._lib.r3dgs_bwd_units_cap(R.capacity, W, H)))
tab = np.zeros((n, 4), np.uint64)
rc = _C._lib.r3dgs_debug_timeline(tab.ctypes.data_as(C.c_void_p), C.c_int(n))
assert rc == 0, rc
tab = tab[tab[:, 1] > 0]   # workgroups beyond the pass's unit count (and tiles nothing contributed to) leave no record
n = len(tab)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "fwd_timeline.npy" if FWD else "bwd_timeline.npy"), tab)
t0, t1, hw, lmax = (tab[:, k].astype(np.int64) for k in range(4))
xcc = (hw >> 32) & 15          # HW_REG_XCC_ID
hw = hw & 0xFFFFFFFF           # HW_REG_HW_ID: wave slot [3:0], simd [5:4], cu [11:8], sh [12], se [15:13]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
dur = (t1 - t0).astype(np.float64)
print(f"{n} workgroups; duration min/median/mean/p90/max = {dur.min():.0f} / {np.median(dur):.0f} / {dur.mean():.0f} / "
      f"{np.percentile(dur, 90):.0f} / {dur.max():.0f} ticks; corr(list length, duration) = {np.corrcoef(lmax, dur)[0, 1]:.2f}")
# s_memtime bases differ between CUs' groups: the timeline is taken per CU and averaged over its own span
groups = {}
for i in range(n):
    groups.setdefault((xcc[i], se[i], sh[i], cu[i]), []).append(i)
sizes = np.array([len(v) for v in groups.values()])
spans, prof = [], np.zeros(10)
for idx in groups.values():
    idx = np.array(idx)
    a, b = t0[idx] - t0[idx].min(), t1[idx] - t0[idx].min()
    span = float(b.max())
    spans.append(span)
    edges = np.linspace(0, span, 11)
    prof += np.array([((a < e1) & (b > e0)).sum() for e0, e1 in zip(edges[:-1], edges[1:])])
spans = np.array(spans)
print(f"{len(groups)} CUs, workgroups per CU {sizes.min()}..{sizes.max()} (mean {sizes.mean():.1f}); per-CU span median "
      f"{np.median(spans):.0f}, p90 {np.percentile(spans, 90):.0f}, max {spans.max():.0f} ticks")
print("mean resident workgroups per CU (4 SIMDs) by tenth of the CU's own span:", np.round(prof / len(groups), 1).tolist())
