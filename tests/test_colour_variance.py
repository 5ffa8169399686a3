"""Next-tier operator `calculate_colours_variance` (reduced_3dgs.cu:41-203).
CPU: the oracle's handling of the reference's tensor-aliasing quirk is cross-checked against the same update written
with real torch aliasing semantics; the product's truncated-colour math runs on the host against the oracle.
GPU (-m gpu): `_C.calculate_colours_variance` through the C ABI vs the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

import synth_scene as ss
from oracle import oracle as orc
from oracle.colour_variance import calculate_colours_variance as oracle_ccv


def _scene(P=1500, ncam=4, W=112, H=80):
    cams = [ss.make_camera(W, H, 80.0, s) for s in [None] + list(range(1, ncam))]
    g = ss.make_gaussians(P, cams[0], seed=21, degree_mode="mixed", scale_mu=0.08)
    args = dict(cam_positions=np.stack([c.camera_center for c in cams]), means3D=g["means3D"], opacity=g["opacity"],
                scales=g["scales"], rotations=g["rotations"],
                cam_viewmatrices=np.stack([c.world_view_transform for c in cams]),
                cam_projmatrices=np.stack([c.full_proj_transform for c in cams]),
                tan_fovxs=np.array([c.tanfovx for c in cams], np.float32),
                tan_fovys=np.array([c.tanfovy for c in cams], np.float32),
                image_height=np.array([H] * ncam, np.int32), image_width=np.array([W] * ncam, np.int32), sh=g["sh"],
                degrees=g["degrees"], max_sh_deg=3)
    return args


def test_oracle_alias_quirk_matches_torch_semantics():
    """reduced_3dgs.cu:185-198: `auto mean_old = mean;` is an alias, so after mean.index_put_ the variance update
    sees the NEW mean in both factors.  Replay one camera's update with real torch aliasing and compare."""
    rng = np.random.default_rng(0)
    P = 64
    mean0 = rng.normal(size=(P, 1, 3)).astype(np.float32)
    var0 = rng.uniform(size=(P, 1, 3)).astype(np.float32)
    colour = rng.uniform(size=(P, 1, 3)).astype(np.float32)
    w = rng.uniform(size=(P, 1)).astype(np.float32)
    wsum = w + rng.uniform(size=(P, 1)).astype(np.float32)
    present = rng.uniform(size=P) < 0.7
    mean, variance = torch.tensor(mean0), torch.tensor(var0)
    col, wt, pres = torch.tensor(colour), torch.tensor(w), torch.tensor(present)
    mean_old = mean  # alias, as in the reference
    coeff = wt / torch.tensor(wsum)
    mean.index_put_((pres,), mean_old[pres] + coeff[pres].view(-1, 1, 1) * (col[pres] - mean_old[pres]))
    variance.index_put_((pres,), variance[pres] + wt[pres].view(-1, 1, 1) * (col[pres] - mean_old[pres]) *
                        (col[pres] - mean[pres]))
    m_new = mean0.copy()
    m_new[present] = mean0[present] + (w / wsum)[present].reshape(-1, 1, 1) * (colour[present] - mean0[present])
    v_new = var0.copy()
    v_new[present] = var0[present] + w[present].reshape(-1, 1, 1) * (colour[present] - m_new[present]) ** 2
    np.testing.assert_allclose(mean.numpy(), m_new, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(variance.numpy(), v_new, rtol=1e-6, atol=1e-7)


def test_truncated_colour_math_on_host_matches_oracle():
    from tests.test_hostcheck import _lib, p
    L = _lib()
    a = _scene(P=700, ncam=1)
    P, M = a["means3D"].shape[0], 16
    deg = np.ascontiguousarray(a["degrees"].reshape(-1))
    campos = np.ascontiguousarray(a["cam_positions"][0], np.float32)
    ref = np.zeros((P, 4, 3), np.float32)
    orc.lib().orc_truncated_colours(C.c_int(P), C.c_int(M), C.c_int(4), p(deg), p(a["means3D"]), p(campos), p(a["sh"]),
                                    p(ref))
    got = np.zeros((P, 4, 3), np.float32)
    L.hc_truncated_colours(C.c_int(P), C.c_int(M), C.c_int(4), p(deg), p(a["means3D"]), p(campos), p(a["sh"]), p(got))
    np.testing.assert_array_equal(got, ref)
    K = (np.arange(4)[None, :] > deg[:, None])
    assert (got[K] == 0).all()  # slots above a Gaussian's own degree stay 0


def test_oracle_outputs_are_sane():
    a = _scene()
    d, v, m = oracle_ccv(**a)
    P = a["means3D"].shape[0]
    assert d.shape == (P, 3) and v.shape == (P, 1, 3) and m.shape == (P, 1, 3)
    seen = ~np.isnan(d[:, 0])
    assert 0.5 < seen.mean() < 1.0               # never-seen Gaussians come back as NaN (0/0), like the reference
    assert (v[seen] >= -1e-7).all() and (d[seen] >= 0).all()
    deg0 = (a["degrees"].reshape(-1) == 0) & seen
    # degree-0 Gaussians: the "full colour" slot [3] is 0, so distance to band 0 is |colour_0| (quirk kept)
    assert (d[deg0, 0] > 0).any()


@pytest.mark.gpu
def test_gpu_matches_oracle():
    from diff_gaussian_rasterization import _C
    a = _scene()
    ref_d, ref_v, ref_m = oracle_ccv(**a)

    def dv(x):
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d, v, m = _C.calculate_colours_variance(dv(a["cam_positions"]), dv(a["means3D"]), dv(a["opacity"]),
                                            dv(a["scales"]), dv(a["rotations"]), dv(a["cam_viewmatrices"]),
                                            dv(a["cam_projmatrices"]), dv(a["tan_fovxs"]), dv(a["tan_fovys"]),
                                            dv(a["image_height"]), dv(a["image_width"]), dv(a["sh"]),
                                            dv(a["degrees"]), 3)
    d, v, m = d.cpu().numpy(), v.cpu().numpy(), m.cpu().numpy()
    assert d.shape == ref_d.shape and v.shape == ref_v.shape and m.shape == ref_m.shape
    np.testing.assert_array_equal(np.isnan(d), np.isnan(ref_d))
    np.testing.assert_array_equal(np.isnan(v), np.isnan(ref_v))
    # The weights are mean transmittances over the pixels a Gaussian was blended into, so ONE threshold-ambiguous pixel
    # (a blend decision the oracle took within 1e-5 of its threshold, which another exp may take the other way) moves a
    # weight by ~1/touched.  Gaussians that can see such a pixel in any camera (pixel within their radius) are compared
    # loosely, all others -- the bulk -- to 1e-5.
    P = a["means3D"].shape[0]
    near_ambiguous = np.zeros(P, bool)
    for i in range(len(a["cam_positions"])):
        H, W = int(a["image_height"][i]), int(a["image_width"][i])
        o = orc.forward(np.zeros(3, np.float32), a["means3D"], None, a["opacity"], a["scales"], a["rotations"], 1.0, None,
                        a["cam_viewmatrices"][i], a["cam_projmatrices"][i], float(a["tan_fovxs"][i]),
                        float(a["tan_fovys"][i]), H, W, a["sh"], a["degrees"], a["cam_positions"][i], want_ambig=True,
                        ambig_rel=1e-5)
        ys, xs = np.nonzero(o["ambig"])
        xy, rad = o["state"]["xy"], o["radii"].astype(np.float32)
        for x, y in zip(xs, ys):
            near_ambiguous |= (rad > 0) & (np.abs(xy[:, 0] - x) <= rad + 1) & (np.abs(xy[:, 1] - y) <= rad + 1)
    seen = ~np.isnan(ref_d[:, 0])
    clean = seen & ~near_ambiguous
    assert clean.sum() >= 0.8 * seen.sum(), (clean.sum(), seen.sum())
    np.testing.assert_allclose(d[clean], ref_d[clean], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v[clean], ref_v[clean], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m[clean], ref_m[clean], rtol=1e-5, atol=1e-6)
    print(f"\ncolour variance: {clean.sum()} of {seen.sum()} seen Gaussians away from ambiguous pixels, max rel err there: "
          f"d {np.abs(d[clean] - ref_d[clean]).max():.2e} v {np.abs(v[clean] - ref_v[clean]).max():.2e} "
          f"m {np.abs(m[clean] - ref_m[clean]).max():.2e}")
    # the others: bounded by one pixel's worth
    np.testing.assert_allclose(d, ref_d, rtol=2e-3, atol=2e-4, equal_nan=True)
    np.testing.assert_allclose(v, ref_v, rtol=5e-3, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(m, ref_m, rtol=2e-3, atol=2e-4)
