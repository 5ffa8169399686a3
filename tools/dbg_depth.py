import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))
import numpy as np, torch
import synth_scene as ss
from diff_gaussian_rasterization import _C
W, H, P = 320, 240, 20000
cam = ss.make_camera(W, H, 250.0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for mode in sys.argv[1:]:
    g = ss.make_gaussians(P, cam, seed=12, degree_mode="mixed", scale_mu=0.02, behind_frac=0.02)
    z = g["means3D"][:, 2]; front = z > 0.2; rng = np.random.default_rng(5)
    if mode == "plane": z[front] = 5.0
    elif mode == "few_depths": z[front] = rng.choice(np.array([2.0, 2.5, 3.0, 4.0, 6.0, 9.0, 11.5], np.float32), int(front.sum()))
    elif mode == "two_far_apart":
        z[front] = (3.0 + 0.01 * rng.random(int(front.sum()))).astype(np.float32); z[np.nonzero(front)[0][:3]] = 90.0
    print("mode", mode, flush=True)
    args = (dev(np.array([.2, .3, .1], np.float32)), dev(g["means3D"]), torch.Tensor([]), dev(g["opacity"]), dev(g["scales"]), dev(g["rotations"]), 1.0,
            torch.Tensor([]), dev(cam.world_view_transform.astype(np.float32)), dev(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, H, W,
            dev(g["sh"]), dev(g["degrees"]), dev(cam.camera_center), False, True)
    out = _C.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    print("  R =", out[0], "img mean", float(out[1].mean()), flush=True)
