"""Generates the committed golden fixtures under tests/golden/.

Run ONLY in the authoring container (needs /root/reference on PYTHONPATH):
    python tests/golden/make_golden.py
Fixtures are data (inputs + expected outputs); no reference source travels.

 ref_sh_eval.npz   : utils/sh_utils.py:57-112 eval_sh outputs for deg 0..3       (reference Python)
 ref_camera.npz    : utils/graphics_utils.py:38-71 getWorld2View2/getProjectionMatrix
                     + the scene/cameras.py:54-58 composition                      (reference Python)
 ref_loss_grad.npz : utils/loss_utils.py:17-66 l1_loss/ssim value and d(loss)/d(image) of
                     0.8*L1 + 0.2*(1-SSIM) (train.py:109-110) -- a realistic dL_dout_color
 ref_cov3d.npz     : utils/general_utils.py:64-110 build_scaling_rotation / strip_symmetric composed as
                     scene/gaussian_model.py:49-54 build_covariance_from_scaling_rotation -- the 6-float 3D covariance the
                     render() path passes as cov3D_precomp when compute_cov3D_python is set            (reference Python)
 ref_pixel_size.npz: scene/__init__.py:103-141 find_minimum_projected_pixel_size_python -- the reference's own torch
                     version of the find_minimum_projected_pixel_size operator, run on CPU      (reference Python)
 ref_sh_backward.npz: autograd through utils/sh_utils.py:57-112 eval_sh exactly as render() composes it
                     (gaussian_renderer/__init__.py:74-81: direction = normalise(xyz - camera_center),
                     clamp_min(eval_sh + 0.5, 0)) -- dL/dsh and the direction part of dL/dmeans for deg 0..3, i.e. the
                     SH backward of backward.cu:20-172 incl. the clamp mask and the normalisation Jacobian  (reference Python)
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


def ref_sh_backward():
    from utils.sh_utils import eval_sh
    rng = np.random.default_rng(23)
    n = 301
    sh = rng.normal(0, 0.4, (n, 16, 3)).astype(np.float32)
    means = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, -4.0], np.float32)
    dL_dcolor = rng.normal(0, 1, (n, 3)).astype(np.float32)
    out = {}
    for deg in range(4):
        K = (deg + 1) ** 2
        sh_t = torch.tensor(sh[:, :K].copy(), requires_grad=True)
        xyz = torch.tensor(means, requires_grad=True)
        shs_view = sh_t.transpose(1, 2).view(-1, 3, K)                       # gaussian_renderer/__init__.py:76
        dir_pp = xyz - torch.tensor(campos).repeat(n, 1)                       # :77
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)          # :78
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)                     # :79
        colors = torch.clamp_min(sh2rgb + 0.5, 0.0)                            # :80
        (colors * torch.tensor(dL_dcolor)).sum().backward()
        out[f"dsh_deg{deg}"] = sh_t.grad.numpy()
        out[f"dmeans_deg{deg}"] = xyz.grad.numpy() if xyz.grad is not None else np.zeros_like(means)  # deg 0: no direction
        out[f"clamped_deg{deg}"] = (sh2rgb.detach().numpy() + 0.5 < 0)
        out[f"margin_deg{deg}"] = np.abs(sh2rgb.detach().numpy() + 0.5)       # distance of each channel from the clamp
    np.savez_compressed(os.path.join(HERE, "ref_sh_backward.npz"), sh=sh, means=means, campos=campos,
                        dL_dcolor=dL_dcolor, **out)


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


def ref_pixel_size():
    """Runs the reference's torch restatement of its pixel-size operator (it hard-codes device="cuda" for two
    helper tensors, so torch.ones is wrapped to drop the device argument; its CUDA-only imports are stubbed)."""
    import types
    for name, attrs in {"simple_knn": [], "simple_knn._C": ["distIndex2", "distCUDA2"],
                        "plyfile": ["PlyData", "PlyElement"],
                        "diff_gaussian_rasterization": ["GaussianRasterizationSettings", "GaussianRasterizer"],
                        "diff_gaussian_rasterization._C": [
                            "calculate_colours_variance", "kmeans_cuda", "sphere_ellipsoid_intersection",
                            "allocate_minimum_redundancy_value", "find_minimum_projected_pixel_size"]}.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    import scene as ref_scene

    cams = []
    for i, (W, H, f) in enumerate([(400, 300, 350.0), (300, 400, 330.0), (640, 360, 500.0), (256, 256, 200.0)]):
        cams.append(ss.make_camera(W, H, f, seed=20 + i))
    g = ss.make_gaussians(3000, ss.make_camera(400, 300, 350.0), seed=5, behind_frac=0.05)
    xyz = g["means3D"]

    class _Cam:
        pass

    tcams = []
    for c in cams:
        t = _Cam()
        t.full_proj_transform = torch.tensor(c.full_proj_transform)
        t.image_width, t.image_height = c.image_width, c.image_height
        tcams.append(t)
    gauss = types.SimpleNamespace(get_opacity=torch.zeros(xyz.shape[0], 1), get_xyz=torch.tensor(xyz),
                                  num_primitives=xyz.shape[0])
    fake = types.SimpleNamespace(gaussians=gauss, getTrainCameras=lambda: tcams)
    real_ones = torch.ones
    torch.ones = lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        sizes = ref_scene.Scene.find_minimum_projected_pixel_size_python(fake)
    finally:
        torch.ones = real_ones
    np.savez_compressed(os.path.join(HERE, "ref_pixel_size.npz"), means3D=xyz,
                        w2ndc=np.stack([c.full_proj_transform for c in cams]),
                        w2ndc_inv=np.stack([torch.tensor(c.full_proj_transform).inverse().numpy() for c in cams]),
                        image_height=np.array([c.image_height for c in cams], np.int32),
                        image_width=np.array([c.image_width for c in cams], np.int32),
                        pixel_sizes=sizes.numpy())


def ref_cov3d():
    """The reference's Python covariance (it hard-codes device="cuda" in torch.zeros: wrapped to drop it)."""
    import utils.general_utils as gu
    rng = np.random.default_rng(17)
    n = 513
    scales = np.exp(rng.normal(-3.0, 1.0, (n, 3))).astype(np.float32)
    rot = rng.normal(0, 1, (n, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    real_zeros = torch.zeros
    torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        out = {}
        for mod in (1.0, 0.6):
            L = gu.build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(rot))
            out[f"cov_mod{mod}"] = gu.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(HERE, "ref_cov3d.npz"), scales=scales, rotations=rot, **out)


if __name__ == "__main__":
    if "--only-cov3d" in sys.argv:
        ref_cov3d()
        sys.exit(0)
    if "--only-pixel-size" in sys.argv:
        ref_pixel_size()
        sys.exit(0)
    if "--only-sh-backward" in sys.argv:
        ref_sh_backward()
        sys.exit(0)
    ref_sh()
    ref_sh_backward()
    ref_cov3d()
    ref_pixel_size()
    ref_camera()
    ref_loss()
    oracle_cases()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
