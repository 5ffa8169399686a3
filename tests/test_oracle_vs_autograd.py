"""C oracle (fp32, analytic backward following DGR/cuda_rasterizer/backward.cu) vs the dense
fp64 PyTorch-autograd restatement (oracle/torch_ref.py).  This is what pins the oracle's
backward: two independent statements of the same maths must agree."""
import os

import numpy as np
import pytest
import torch

import synth_scene as ss
from oracle import oracle as orc
from oracle import torch_ref as tr

DT = torch.float64


def T(a):
    return torch.tensor(np.asarray(a), dtype=DT)


def run_case(P, W, H, f, cam_seed, gseed, degree_mode, bg, lam=0.0, precomp_color=False, precomp_cov=False,
             spread=1.0, dl=None, mod=1.0):
    cam = ss.make_camera(W, H, f, cam_seed)
    g = ss.make_gaussians(P, cam, seed=gseed, degree_mode=degree_mode, scale_mu=0.15, scale_sigma=0.7)
    g["means3D"][:, :2] *= spread  # >1 pushes Gaussians beyond 1.3*tanfov => exercises the EWA clamp
    bg = np.array(bg, np.float32)
    colors = cov = None
    rng = np.random.default_rng(gseed + 77)
    if precomp_color:
        colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    if precomp_cov:
        Rm = tr.quat_to_R(torch.tensor(g["rotations"], dtype=DT)).numpy()
        L = Rm * g["scales"][:, None, :].astype(np.float64)
        S = L @ L.transpose(0, 2, 1)
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    out = orc.forward(bg, g["means3D"], colors, g["opacity"], None if precomp_cov else g["scales"],
                      None if precomp_cov else g["rotations"], mod, cov, cam.world_view_transform,
                      cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, None if precomp_color else g["sh"],
                      g["degrees"], cam.camera_center, want_ambig=True)
    leaves = dict(m3=T(g["means3D"]), op=T(g["opacity"]), sc=T(g["scales"]), rot=T(g["rotations"]), sh=T(g["sh"]))
    if precomp_color:
        leaves["col"] = T(colors)
    if precomp_cov:
        leaves["cov"] = T(cov)
    for v in leaves.values():
        v.requires_grad_()
    col, radii, sp = tr.render(leaves["m3"], leaves["op"], leaves["sc"], leaves["rot"], leaves["sh"],
                               torch.tensor(g["degrees"]), T(cam.world_view_transform),
                               T(cam.full_proj_transform), T(cam.camera_center), T(bg), W, H, cam.tanfovx,
                               cam.tanfovy, scale_modifier=mod, colors_precomp=leaves.get("col"),
                               cov3D_precomp=leaves.get("cov"), lambda_sh_sparsity=lam)
    if dl is None:
        dl = ss.upstream_grad(W, H, seed=gseed + 3) * W * H
    ((col * T(dl)).sum() + sp).backward()
    gr = orc.backward(out["state"], dl, lam)
    return out, col.detach().numpy(), radii.numpy(), leaves, gr


def close(name, ref, got, rel=3e-4):
    """fp32 oracle vs fp64 autograd.  The bound is relative to the tensor's max |grad|: per-term fp32
    rounding (1-alpha near 0.99, T recovered by division, backward.cu:541) is ~1e-6 and the zero-mean
    upstream gradient makes per-Gaussian sums cancel ~30-50x, so ~1e-4 is the fp32 noise floor here;
    typical agreement is 1e-6 (see the 'mixed' case)."""
    ref = ref.numpy() if hasattr(ref, "numpy") else ref
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(ref - got.reshape(ref.shape)).max()
    assert err <= rel * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("kw", [
    dict(P=400, W=56, H=40, f=45.0, cam_seed=3, gseed=5, degree_mode="mixed", bg=(0.3, 0.6, 0.1)),
    dict(P=300, W=48, H=48, f=40.0, cam_seed=None, gseed=1, degree_mode="all3", bg=(0, 0, 0)),
    dict(P=300, W=33, H=47, f=40.0, cam_seed=6, gseed=2, degree_mode="all0", bg=(1, 1, 1)),
    dict(P=350, W=56, H=40, f=45.0, cam_seed=7, gseed=8, degree_mode="mixed", bg=(0.5, 0.5, 0.5), lam=0.1),
    dict(P=350, W=56, H=40, f=45.0, cam_seed=8, gseed=9, degree_mode="all3", bg=(0.1, 0.2, 0.3), spread=1.35),
], ids=["mixed", "deg3_black_idcam", "deg0_white_ragged_edges", "sh_sparsity", "ewa_clamp"])
def test_forward_and_backward_match_autograd(kw):
    out, col, radii, lv, gr = run_case(**kw)
    np.testing.assert_array_equal(out["radii"], radii)
    ok = out["ambig"] == 0
    assert ok.mean() > 0.99
    assert np.abs(col - out["color"])[:, ok].max() < 5e-6
    close("means3D", lv["m3"].grad, gr["dL_dmeans3D"])
    close("opacity", lv["op"].grad, gr["dL_dopacity"])
    close("scales", lv["sc"].grad, gr["dL_dscales"])
    close("rotations", lv["rot"].grad, gr["dL_drotations"])
    close("sh", lv["sh"].grad, gr["dL_dsh"])
    # API contract: dL_dsh beyond the Gaussian's own band count is exactly 0; culled Gaussians get 0
    K = (np.arange(16)[None, :] >= ((out["state"]["degrees"].reshape(-1) + 1) ** 2)[:, None])
    assert (gr["dL_dsh"][K] == 0).all()
    inv = out["radii"] == 0
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dmeans2D"):
        assert (gr[k][inv] == 0).all()


def test_precomputed_colour_and_covariance_paths():
    out, col, radii, lv, gr = run_case(P=300, W=48, H=40, f=40.0, cam_seed=5, gseed=4, degree_mode="all3",
                                       bg=(0.2, 0.2, 0.9), precomp_color=True, precomp_cov=True)
    np.testing.assert_array_equal(out["radii"], radii)
    ok = out["ambig"] == 0
    assert np.abs(col - out["color"])[:, ok].max() < 5e-6
    close("colors_precomp", lv["col"].grad, gr["dL_dcolors"])
    close("cov3D_precomp", lv["cov"].grad, gr["dL_dcov3D"])
    close("means3D", lv["m3"].grad, gr["dL_dmeans3D"])
    assert (gr["dL_dscales"] == 0).all() and (gr["dL_drotations"] == 0).all()


def test_realistic_upstream_gradient_from_reference_loss(golden_dir):
    """dL_dout_color = gradient of 0.8*L1 + 0.2*(1-SSIM) computed by the reference's loss_utils."""
    dl = np.load(os.path.join(golden_dir, "ref_loss_grad.npz"))["dloss_dimage"]
    out, col, radii, lv, gr = run_case(P=400, W=56, H=40, f=45.0, cam_seed=3, gseed=5, degree_mode="mixed",
                                       bg=(0, 0, 0), dl=dl)
    close("means3D", lv["m3"].grad, gr["dL_dmeans3D"])
    close("sh", lv["sh"].grad, gr["dL_dsh"])
    close("rotations", lv["rot"].grad, gr["dL_drotations"])


def test_counter_mode_and_mark_visible():
    """forward.cu:560-564 counters: touched = #pixels blended, transmittance = sum of T before blending."""
    cam = ss.make_camera(48, 40, 40.0, 2)
    g = ss.make_gaussians(200, cam, seed=3, degree_mode="all0", scale_mu=0.15)
    out = orc.forward(np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0,
                      None, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 40, 48,
                      g["sh"], g["degrees"], cam.camera_center, counter_mode=True)
    assert out["touched_pixels"].sum() > 0
    assert (out["touched_pixels"][out["radii"] == 0] == 0).all()
    assert (out["transmittance"] <= out["touched_pixels"] + 1e-4).all()  # T <= 1 per blended pixel
    # sum over Gaussians of touched == sum over pixels of #blended entries <= n_contrib
    assert out["touched_pixels"].sum() <= out["state"]["n_contrib"].sum()
    vis = orc.mark_visible(g["means3D"], cam.world_view_transform)
    t = np.concatenate([g["means3D"], np.ones((200, 1), np.float32)], 1) @ cam.world_view_transform
    np.testing.assert_array_equal(vis, t[:, 2] > 0.2)
    assert (vis[out["radii"] > 0]).all()


def test_ragged_sh_matches_dense():
    """forward.cu:19-36,245-350: degree-sorted ragged SH buffer gives the same image as dense SH + degrees."""
    cam = ss.make_camera(48, 40, 40.0, 1)
    g = ss.make_gaussians(300, cam, seed=6, degree_mode="mixed", scale_mu=0.15)
    order = np.argsort(g["degrees"].reshape(-1), kind="stable")
    g = {k: v[order] for k, v in g.items()}
    deg = g["degrees"].reshape(-1)
    per_band = np.array([(deg == d).sum() for d in range(4)], np.int32)
    cumsum = np.cumsum(per_band).astype(np.int32)
    coeffs = np.array([1, 4, 9, 16], np.int32)
    flat = np.concatenate([g["sh"][deg == d][:, :(d + 1) ** 2].reshape(-1) for d in range(4)]).astype(np.float32)
    args = (np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
            cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 40, 48)
    dense = orc.forward(*args, g["sh"], g["degrees"], cam.camera_center)
    rag = orc.forward(*args, flat, g["degrees"], cam.camera_center, ragged=(coeffs, per_band, cumsum))
    np.testing.assert_array_equal(dense["color"], rag["color"])
    np.testing.assert_array_equal(dense["radii"], rag["radii"])


def test_empty_and_all_culled_inputs():
    cam = ss.make_camera(32, 32, 30.0, None)
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    e = np.zeros((0, 3), np.float32)
    out = orc.forward(bg, e, None, np.zeros((0, 1), np.float32), e, np.zeros((0, 4), np.float32), 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 32, 32,
                      np.zeros((0, 16, 3), np.float32), np.zeros((0, 1), np.int32), cam.camera_center)
    assert out["num_rendered"] == 0 and (out["color"] == 0).all()  # reference: P==0 leaves the zero image
    g = ss.make_gaussians(50, cam, seed=0, degree_mode="all0")
    g["means3D"][:, 2] = -1.0  # all behind the camera
    out = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 32, 32,
                      g["sh"], g["degrees"], cam.camera_center)
    assert out["num_rendered"] == 0 and (out["radii"] == 0).all()
    np.testing.assert_array_equal(out["color"], np.broadcast_to(bg[:, None, None], (3, 32, 32)))


def test_tiled_torch_reference_equals_dense_and_matches_oracle():
    """oracle/torch_ref.py render(tiled=True) -- the configs[0] "PyTorch CPU autograd reference render" bench.py times as
    cpu_baseline -- is the dense statement evaluated tile by tile: identical image and gradients, and the C oracle's
    image within fp32 rounding."""
    W, H, P = 70, 45, 400
    cam = ss.make_camera(W, H, 60.0, 3)
    g = ss.make_gaussians(P, cam, seed=9, degree_mode="mixed", scale_mu=0.12, scale_sigma=0.5)
    bg = np.array([0.2, 0.1, 0.4], np.float32)
    dl = torch.tensor(ss.upstream_grad(W, H, seed=4) * (W * H), dtype=DT)

    def run(tiled):
        leaves = dict(m3=T(g["means3D"]), op=T(g["opacity"]), sc=T(g["scales"]), rot=T(g["rotations"]), sh=T(g["sh"]))
        for v in leaves.values():
            v.requires_grad_()
        color, radii, _ = tr.render(leaves["m3"], leaves["op"], leaves["sc"], leaves["rot"], leaves["sh"],
                                    torch.tensor(g["degrees"]), T(cam.world_view_transform), T(cam.full_proj_transform),
                                    T(cam.camera_center), T(bg), W, H, cam.tanfovx, cam.tanfovy, tiled=tiled)
        (color * dl).sum().backward()
        return color.detach(), radii, {k: v.grad.clone() for k, v in leaves.items()}

    c0, r0, g0 = run(False)
    c1, r1, g1 = run(True)
    assert torch.equal(r0, r1)
    assert float((c0 - c1).abs().max()) < 1e-12
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-10 * (float(g0[k].abs().max()) + 1e-30), k
    ref = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center, want_ambig=True)
    ok = ref["ambig"].reshape(-1) == 0
    assert np.abs(c1.numpy().reshape(3, -1) - ref["color"].reshape(3, -1))[:, ok].max() < 2e-5
