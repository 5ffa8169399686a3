"""Generates the committed golden fixtures under tests/golden/.

Run ONLY in the authoring container (needs /root/reference on PYTHONPATH):
    python tests/golden/make_golden.py
Fixtures are data (inputs + expected outputs); no reference source travels.

 ref_sh_eval.npz   : utils/sh_utils.py:57-112 eval_sh outputs for deg 0..3       (reference Python)
 ref_camera.npz    : utils/graphics_utils.py:38-71 getWorld2View2/getProjectionMatrix
                     + the scene/cameras.py:54-58 composition                      (reference Python)
 ref_loss_grad.npz : utils/loss_utils.py:17-66 l1_loss/ssim value and d(loss)/d(image) of
                     0.8*L1 + 0.2*(1-SSIM) (train.py:109-110) -- a realistic dL_dout_color
 oracle_case_*.npz : outputs of the repo's own CPU oracle on small seeded scenes (regression
                     anchors + GPU parity targets that do not need a compiler on the GPU box)
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import synth_scene as ss  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.golden_cases import CASES, case_inputs  # noqa: E402


def ref_sh():
    from utils.sh_utils import eval_sh
    rng = np.random.default_rng(7)
    n = 257
    sh = rng.normal(0, 0.4, (n, 16, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    out = {}
    for deg in range(4):
        K = (deg + 1) ** 2
        shs_view = torch.tensor(sh[:, :K]).transpose(1, 2)  # [n,3,K] as gaussian_renderer/__init__.py:76
        out[f"rgb_deg{deg}"] = eval_sh(deg, shs_view, torch.tensor(d)).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_sh_eval.npz"), sh=sh, dirs=d, **out)


def ref_camera():
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    rng = np.random.default_rng(11)
    cases = {}
    for i in range(4):
        ang = rng.uniform(-0.6, 0.6, 3)
        R = ss.rot_xyz(*ang)
        T = rng.uniform(-2, 2, 3)
        fovx, fovy = rng.uniform(0.5, 1.4), rng.uniform(0.4, 1.2)
        w2v = getWorld2View2(R, T)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
        wvt = torch.tensor(w2v).transpose(0, 1)
        pm = proj.transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cases.update({f"R{i}": R, f"T{i}": T, f"fov{i}": np.array([fovx, fovy]), f"w2v{i}": w2v,
                      f"proj{i}": proj.numpy(), f"full{i}": full.numpy(), f"center{i}": center.numpy()})
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **cases)


def ref_loss():
    from utils.loss_utils import l1_loss, ssim
    g = torch.Generator().manual_seed(1)
    img = torch.rand(3, 40, 56, generator=g).requires_grad_()
    gt = torch.rand(3, 40, 56, generator=g)
    Ll1 = l1_loss(img, gt)
    s = ssim(img, gt)
    loss = 0.8 * Ll1 + 0.2 * (1.0 - s)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "ref_loss_grad.npz"), image=img.detach().numpy(), gt=gt.numpy(),
                        l1=Ll1.item(), ssim=s.item(), loss=loss.item(), dloss_dimage=img.grad.numpy())


def oracle_cases():
    for name, kw in CASES.items():
        cam, g, bg, dl = case_inputs(kw)
        out = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                          cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                          kw["H"], kw["W"], g["sh"], g["degrees"], cam.camera_center, want_ambig=True)
        st = out["state"]
        gr = orc.backward(st, dl, kw["lam"])
        np.savez_compressed(
            os.path.join(HERE, f"oracle_case_{name}.npz"),
            color=out["color"], radii=out["radii"], num_rendered=out["num_rendered"], ambig=out["ambig"],
            keys=st["keys"], point_list=st["point_list"], ranges=st["ranges"], n_contrib=st["n_contrib"],
            final_T=st["final_T"], **{k: v for k, v in gr.items()})


if __name__ == "__main__":
    ref_sh()
    ref_camera()
    ref_loss()
    oracle_cases()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
