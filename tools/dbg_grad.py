import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reduced-3dgs_amd")]
import synth_scene as ss
from oracle import oracle as orc
from tests import test_gpu_parity as t
from diff_gaussian_rasterization import _C
W, H, P = 1600, 1062, 500_000
cam = ss.make_camera(W, H, 1200.0, None)
g = ss.make_gaussians(P, cam, seed=0, degree_mode="all3", scale_mu=0.012)
bg = np.array([0.1, 0.4, 0.9], np.float32)
dl = ss.upstream_grad(W, H, seed=2) * (W * H)
ref = t.oracle_forward(bg, g, cam, H, W)
amb = ref["ambig"].reshape(-1) != 0
print("ambiguous pixels", amb.sum(), "of", amb.size)
for mask in (False, True):
    d = dl.copy()
    if mask:
        d.reshape(3, -1)[:, amb] = 0
    gr = orc.backward(ref["state"], d, 0.0)
    for tight in (False, True):
        _C.set_tight_rects(tight)
        fargs, fout = t.hip_forward(_C, bg, g, cam, H, W, exact=True)
        b = t.hip_backward(_C, fargs, fout, d, 0.0)
        names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
        out = []
        for n, got in zip(names, b[:8]):
            r = gr[n]
            e = np.abs(got.cpu().numpy().reshape(r.shape) - r)
            scale = np.abs(r).max()
            pe = e <= 1e-4 * np.abs(r) + 1e-6 * scale
            out.append(f"{n[3:]}: max {e.max()/scale:.1e} per-elem-ok {pe.mean():.5f}")
        print("masked" if mask else "unmasked", "tight" if tight else "ref", " | ".join(out))
_C.set_tight_rects(True)
