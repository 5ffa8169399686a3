"""Debug aid: the backward's unit lists (count and segment length per list, weight-class profile) of one workload, for the
library selected by R3DGS_LIB -- to compare two builds' launch orders."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))
import synth_scene as ss
from diff_gaussian_rasterization import _C
name = sys.argv[1] if len(sys.argv) > 1 else "garden_like_2M_1600x1062"
w, cam, g = ss.make_workload(name)
W, H, P = w["W"], w["H"], w["P"]
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = (dv(np.zeros(3, np.float32)), dv(g["means3D"]), torch.empty(0), dv(g["opacity"]), dv(g["scales"]), dv(g["rotations"]), 1.0,
        torch.empty(0), dv(cam.world_view_transform), dv(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, H, W, dv(g["sh"]),
        dv(g["degrees"]), dv(cam.camera_center), False, False)
for it in range(2):
    out = _C.rasterize_gaussians(*args)
R, color, radii, geom, binning, img = out
dl = dv(ss.upstream_grad(W, H, seed=1) * (W * H))
(bg, m3, colors, op, sc, rot, mod, cov, vm, pm, tx, ty, H_, W_, sh, deg, campos, _, _) = args
_C.rasterize_gaussians_backward(bg, m3, radii, colors, sc, rot, mod, cov, vm, pm, tx, ty, dl, sh, deg, campos, geom, R, binning, img, 0.0, False)
torch.cuda.synchronize()
ex = _C.export_tile_order(H, W, img, P=P, num_rendered=R, binningBuffer=binning)
print(_C.LIBRARY_PATH, "pairs", R.pairs)
for i, l in enumerate(ex["lists"]):
    segs = l["segments"]
    print(f" list {i}: units {len(l['tile'])}, walk {l['walk']}, tiles split {(segs > 1).sum().item() if hasattr(segs, 'sum') else 0}, first 6 units (tile, seg, nseg) {[(int(a), int(b), int(c)) for a, b, c in zip(l['tile'][:6], l['segment'][:6], l['segments'][:6])]}")
