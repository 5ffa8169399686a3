"""oracle/backward_f64.c -- the double-precision, independently derived backward -- pinned against fp64 autograd through
oracle/torch_ref.py, and the fp32 oracle (raster_oracle.c, the transcription of backward.cu's arithmetic) measured
against it.

Why it exists: the GPU kernels and raster_oracle.c evaluate the covariance chain (backward.cu:228-306,
311-374) in fp32 from the same closed forms; when the two differ by 1.2e-4 of the tensor's maximum nothing says which one
is off.  backward_f64 is the exact gradient of the function the forward evaluated; the GPU tests compare BOTH fp32
evaluations with it."""
import numpy as np
import pytest
import torch

import synth_scene as ss
from oracle import oracle as orc
from oracle import torch_ref as tr

DT = torch.float64


def T(a):
    return torch.tensor(np.asarray(a), dtype=DT)


def scene(P, W, H, f, cam_seed, gseed, degree_mode, scale_mu=0.15, scale_sigma=0.7, spread=1.0):
    cam = ss.make_camera(W, H, f, cam_seed)
    g = ss.make_gaussians(P, cam, seed=gseed, degree_mode=degree_mode, scale_mu=scale_mu, scale_sigma=scale_sigma)
    g["means3D"][:, :2] *= spread
    return cam, g


def oracle_fwd(cam, g, bg, W, H, ambig_rel=1e-4, mod=1.0):
    return orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], mod, None,
                       cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                       g["degrees"], cam.camera_center, want_ambig=True, ambig_rel=ambig_rel)


def f32(v):
    """tan(fov/2) as the C ABI receives it (a float argument)"""
    return float(np.float32(v))


def relerr(ref, got):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(ref - np.asarray(got, np.float64).reshape(ref.shape)).max() / (np.abs(ref).max() + 1e-300))


@pytest.mark.parametrize("kw", [
    dict(P=400, W=56, H=40, f=45.0, cam_seed=3, gseed=5, degree_mode="mixed", bg=(0.3, 0.6, 0.1), lam=0.0),
    dict(P=300, W=48, H=48, f=40.0, cam_seed=None, gseed=1, degree_mode="all3", bg=(0, 0, 0), lam=0.0),
    dict(P=350, W=56, H=40, f=45.0, cam_seed=7, gseed=8, degree_mode="mixed", bg=(0.5, 0.5, 0.5), lam=0.1),
    dict(P=350, W=56, H=40, f=45.0, cam_seed=8, gseed=9, degree_mode="all3", bg=(0.1, 0.2, 0.3), lam=0.0, spread=1.35),
    dict(P=300, W=48, H=40, f=40.0, cam_seed=2, gseed=3, degree_mode="all3", bg=(0.9, 0.2, 0.3), lam=0.0, mod=1.3),
], ids=["mixed", "deg3_black_idcam", "sh_sparsity", "ewa_clamp", "scale_modifier"])
def test_f64_backward_is_the_autograd_gradient(kw):
    """Fed the fp64 forward's own per-Gaussian numbers (means, conic, opacity, colour, covariance), and with the pixels
    whose skip decisions fp32 and fp64 may take differently removed from the upstream gradient, the hand-derived double
    backward must BE the autograd gradient: 1e-10 of each tensor's maximum in its `pure` mode.  With the reference's
    conventions (the default: focal lengths and clamp limits formed in fp32, and backward.cu:234's 1 / (det^2 + 1e-7), which
    is not the derivative of the forward's 1 / det) it stays within 1e-6 of it."""
    W, H, P, mod = kw["W"], kw["H"], kw["P"], f32(kw.get("mod", 1.0))
    cam, g = scene(P, W, H, kw["f"], kw["cam_seed"], kw["gseed"], kw["degree_mode"], spread=kw.get("spread", 1.0))
    bg = np.array(kw["bg"], np.float32)
    out = oracle_fwd(cam, g, bg, W, H, mod=mod)
    dl = ss.upstream_grad(W, H, seed=kw["gseed"] + 3) * W * H
    dl.reshape(3, -1)[:, out["ambig"].reshape(-1) != 0] = 0.0
    lv = dict(m3=T(g["means3D"]), op=T(g["opacity"]), sc=T(g["scales"]), rot=T(g["rotations"]), sh=T(g["sh"]))
    for v in lv.values():
        v.requires_grad_()
    geo = {}
    col, radii, sp = tr.render(lv["m3"], lv["op"], lv["sc"], lv["rot"], lv["sh"], torch.tensor(g["degrees"]),
                               T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center), T(bg), W, H,
                               f32(cam.tanfovx), f32(cam.tanfovy), scale_modifier=mod, lambda_sh_sparsity=kw["lam"],
                               geo_out=geo)
    np.testing.assert_array_equal(out["radii"], radii.numpy())
    ((col * T(dl)).sum() + sp).backward()
    for pure, bar in ((True, 1e-10), (False, 1e-6)):
        g64 = orc.backward_f64(out["state"], dl, kw["lam"], fwd64=geo, pure=pure)
        # dL_dscales is the gradient w.r.t. the modifier-scaled scale (backward.cu:355-358): autograd's is mod times that
        for name, ref, got in (("means3D", lv["m3"].grad, g64["dL_dmeans3D"]), ("opacity", lv["op"].grad, g64["dL_dopacity"]),
                               ("scales", lv["sc"].grad / mod, g64["dL_dscales"]),
                               ("rotations", lv["rot"].grad, g64["dL_drotations"]), ("sh", lv["sh"].grad, g64["dL_dsh"])):
            e = relerr(ref.numpy(), got)
            assert e <= bar, f"{name} (pure={pure}): {e:.2e}"


def test_f64_backward_precomputed_covariance_and_colour():
    """dL_dcov3D and dL_dcolors have no leaf on the SH / scale+rotation route; check them on the precomputed route."""
    W, H, P = 48, 40, 300
    cam, g = scene(P, W, H, 40.0, 5, 4, "all3")
    rng = np.random.default_rng(81)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    Rm = tr.quat_to_R(T(g["rotations"])).numpy()
    L = Rm * g["scales"][:, None, :].astype(np.float64)
    S = L @ L.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    bg = np.array([0.2, 0.2, 0.9], np.float32)
    out = orc.forward(bg, g["means3D"], colors, g["opacity"], None, None, 1.0, cov, cam.world_view_transform,
                      cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, None, g["degrees"], cam.camera_center,
                      want_ambig=True)
    dl = ss.upstream_grad(W, H, seed=7) * W * H
    dl.reshape(3, -1)[:, out["ambig"].reshape(-1) != 0] = 0.0
    lv = dict(m3=T(g["means3D"]), op=T(g["opacity"]), col=T(colors), cov=T(cov))
    for v in lv.values():
        v.requires_grad_()
    geo = {}
    col, radii, _ = tr.render(lv["m3"], lv["op"], T(g["scales"]), T(g["rotations"]), T(g["sh"]), torch.tensor(g["degrees"]),
                              T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center), T(bg), W, H,
                              f32(cam.tanfovx), f32(cam.tanfovy), colors_precomp=lv["col"], cov3D_precomp=lv["cov"],
                              geo_out=geo)
    (col * T(dl)).sum().backward()
    g64 = orc.backward_f64(out["state"], dl, 0.0, fwd64=geo, pure=True)
    for name, ref, got in (("colors", lv["col"].grad, g64["dL_dcolors"]), ("cov3D", lv["cov"].grad, g64["dL_dcov3D"]),
                           ("means3D", lv["m3"].grad, g64["dL_dmeans3D"]), ("opacity", lv["op"].grad, g64["dL_dopacity"])):
        e = relerr(ref.numpy(), got)
        assert e <= 1e-10, f"{name}: {e:.2e}"
    assert (g64["dL_dscales"] == 0).all() and (g64["dL_drotations"] == 0).all()


@pytest.mark.parametrize("kw", [
    dict(P=10_000, W=400, H=400, f=300.0, cam_seed=None, gseed=0, degree_mode="all0", scale_mu=0.012, lam=0.0),
    dict(P=20_000, W=640, H=360, f=400.0, cam_seed=3, gseed=4, degree_mode="mixed", scale_mu=0.02, lam=0.1),
], ids=["cfg0_10k_400x400", "20k_640x360_mixed"])
def test_fp32_oracle_against_f64(kw):
    """What the reference's fp32 arithmetic (as restated by raster_oracle.c, per-Gaussian sums in double) is worth against
    the exact gradient, at BASELINE.json configs[0] and a mixed-degree scene (printed with -s; the GPU suite prints the same
    numbers for the HIP kernels at every size)."""
    W, H, P = kw["W"], kw["H"], kw["P"]
    cam = ss.make_camera(W, H, kw["f"], kw["cam_seed"])
    g = ss.make_gaussians(P, cam, seed=kw["gseed"], degree_mode=kw["degree_mode"], scale_mu=kw["scale_mu"])
    bg = np.array([0.1, 0.4, 0.9], np.float32)
    out = oracle_fwd(cam, g, bg, W, H, ambig_rel=1e-5)
    dl = ss.upstream_grad(W, H, seed=2) * (W * H)
    dl.reshape(3, -1)[:, out["ambig"].reshape(-1) != 0] = 0.0
    g32 = orc.backward(out["state"], dl, kw["lam"])
    g64 = orc.backward_f64(out["state"], dl, kw["lam"])
    errs = {k: relerr(g64[k], g32[k]) for k in ("dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_dopacity", "dL_dmeans3D",
                                                 "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")}
    print("fp32 oracle vs f64:", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 1e-4, f"{k}: {v:.2e}"


def test_configs0_three_parties_fp64_autograd_f64_oracle_fp32_oracle():
    """BASELINE.json configs[0] (10k Gaussians, 400x400, degree 0) by three independent parties: the tile-by-tile fp64
    autograd statement (oracle/torch_ref.py -- the only statement of the EWA projection / blend maths that shares no
    expression with the product), the C oracle's fp32 forward + backward (raster_oracle.c), and the double backward
    (backward_f64.c).  Radii equal; colour within 2e-5 (5e-6 on 99.5 %) of the pixels that are not threshold-ambiguous; the fp32 oracle's
    gradients within 3e-4 of autograd's (relative to each tensor's maximum); the double backward's within 1e-9 when it is fed
    the fp64 forward's per-Gaussian numbers (same function, two derivations) and within 1e-4 on the fp32 forward's state
    (measured ~1e-5: what rounding the stored pixel means / conics to fp32 is worth once the zero-mean upstream gradient
    has cancelled the sums -- a property of the reference's fp32 forward buffers, the same for every backward)."""
    w = ss.WORKLOADS["cfg0_10k_400"]
    W, H, P = w["W"], w["H"], w["P"]
    cam = ss.make_camera(W, H, w["f"], None)
    g = ss.make_gaussians(P, cam, seed=0, degree_mode=w["degree_mode"])
    bg = np.array([0.1, 0.4, 0.9], np.float32)
    out = oracle_fwd(cam, g, bg, W, H, ambig_rel=1e-4)
    ok = out["ambig"].reshape(-1) == 0
    assert ok.mean() > 0.995
    dl = ss.upstream_grad(W, H, seed=2) * (W * H)
    dl.reshape(3, -1)[:, ~ok] = 0.0
    lv = dict(m3=T(g["means3D"]), op=T(g["opacity"]), sc=T(g["scales"]), rot=T(g["rotations"]), sh=T(g["sh"]))
    for v in lv.values():
        v.requires_grad_()
    geo = {}
    col, radii, _ = tr.render(lv["m3"], lv["op"], lv["sc"], lv["rot"], lv["sh"], torch.tensor(g["degrees"]),
                              T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center), T(bg), W, H,
                              f32(cam.tanfovx), f32(cam.tanfovy), tiled=True, geo_out=geo)
    np.testing.assert_array_equal(out["radii"], radii.numpy())
    cerr = np.abs(col.detach().numpy().reshape(3, -1) - out["color"].reshape(3, -1))[:, ok].max()
    assert cerr <= 2e-5, cerr   # fp32 rounding of the oracle's forward over ~30 blended entries (measured 1.3e-5), no flipped decision
    assert (np.abs(col.detach().numpy().reshape(3, -1) - out["color"].reshape(3, -1))[:, ok].max(0) > 5e-6).mean() < 5e-3
    (col * T(dl)).sum().backward()
    g32 = orc.backward(out["state"], dl, 0.0)
    g64 = orc.backward_f64(out["state"], dl, 0.0)
    g64p = orc.backward_f64(out["state"], dl, 0.0, fwd64=geo, pure=True)
    rows = []
    for name, ref, key in (("means3D", lv["m3"].grad, "dL_dmeans3D"), ("opacity", lv["op"].grad, "dL_dopacity"),
                           ("scales", lv["sc"].grad, "dL_dscales"), ("rotations", lv["rot"].grad, "dL_drotations"),
                           ("sh", lv["sh"].grad, "dL_dsh")):
        e32, e64, e64p = relerr(ref.numpy(), g32[key]), relerr(ref.numpy(), g64[key]), relerr(ref.numpy(), g64p[key])
        rows.append(f"{name} fp32 {e32:.1e} f64-on-fp32-state {e64:.1e} f64-on-fp64-state {e64p:.1e}")
        assert e32 <= 3e-4, f"{name}: fp32 oracle vs autograd {e32:.2e}"
        assert e64 <= 1e-4, f"{name}: f64 oracle on the fp32 forward state vs autograd {e64:.2e}"
        assert e64p <= 1e-9, f"{name}: f64 oracle on the fp64 forward state vs autograd {e64p:.2e}"
    print("configs[0] vs fp64 autograd:", f"colour {cerr:.1e};", "; ".join(rows))
