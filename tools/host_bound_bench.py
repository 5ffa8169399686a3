"""Host-bound shapes: training steps per second of the 10k-Gaussian configs[0] workload (GPU stages sum to ~0.13 ms, the
rest of a step is the host layer) through the compiled torch binding and through the ctypes marshalling, strict mode on.

    python tools/host_bound_bench.py [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reduced-3dgs_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth_scene as ss  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import _C  # noqa: E402


def run(workload, binding, steps):
    w, cam, g = ss.make_workload(workload)
    W, H = w["W"], w["H"]
    dev = torch.device("cuda", 0)

    def dv(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    leaves = {k: dv(g[k]).requires_grad_() for k in ("means3D", "opacity", "scales", "rotations", "sh")}
    degrees, bg, dl, empty = dv(g["degrees"]), dv(np.zeros(3, np.float32)), dv(ss.upstream_grad(W, H, seed=1)), torch.Tensor([])
    rs = dgr.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, dv(cam.world_view_transform),
                                           dv(cam.full_proj_transform), 3, dv(cam.camera_center), False, False)
    torch.autograd.set_multithreading_enabled(False)
    _C.set_binding(binding)

    def step():
        for t in leaves.values():
            t.grad = None
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0
        means2D.retain_grad()
        color, radii = dgr.rasterize_gaussians(leaves["means3D"], means2D, leaves["sh"], degrees, empty, leaves["opacity"],
                                               leaves["scales"], leaves["rotations"], empty, rs, 0.0)
        color.backward(dl)
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return steps / best


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    out = {}
    for workload in ("cfg0_10k_400", "lego_like_300k_800"):
        for binding in ("ctypes", "torch"):
            out[f"{workload}/{binding}"] = round(run(workload, binding, steps), 1)
    out["strict"] = _C.is_strict()
    print(json.dumps(out))
