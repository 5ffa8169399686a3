"""Seeded synthetic scenes + cameras for tests, smoke() and bench.py (SURVEY.md 8d recipe).

Test/bench infrastructure -- not part of the product package.  Camera matrices follow the
reference's conventions: `world_view_transform = getWorld2View2(R, T)^T`,
`projection = getProjectionMatrix(0.01, 100, FoVx, FoVy)^T`, `full_proj = view @ proj`,
`camera_center = inverse(view)[3, :3]` (scene/cameras.py:48-58, utils/graphics_utils.py:38-71).
The two matrix builders below are restated from those formulas and pinned against the
reference functions by tests/golden/camera_*.npz (tests/test_oracle_pins.py).
"""
import math

import numpy as np


def world2view(R, t):
    """utils/graphics_utils.py:38-50 getWorld2View2 with translate=0, scale=1.
    R is the camera-to-world rotation as stored by the reference (it transposes it)."""
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = np.asarray(R, np.float64).T
    Rt[:3, 3] = np.asarray(t, np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:51-71 getProjectionMatrix (float32 like the torch.zeros(4,4) original)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    Pm = np.zeros((4, 4), np.float32)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


class Camera:
    """Minimal stand-in for scene/cameras.py:Camera exposing the attributes render() reads."""

    def __init__(self, W, H, fx, fy, R=None, T=None):
        self.image_width, self.image_height = int(W), int(H)
        self.FoVx = 2 * math.atan(W / (2 * fx))
        self.FoVy = 2 * math.atan(H / (2 * fy))
        self.R = np.eye(3) if R is None else np.asarray(R, np.float64)
        self.T = np.zeros(3) if T is None else np.asarray(T, np.float64)
        self.znear, self.zfar = 0.01, 100.0
        self.world_view_transform = world2view(self.R, self.T).T.copy()
        self.projection_matrix = projection(self.znear, self.zfar, self.FoVx, self.FoVy).T.copy()
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).astype(np.float32)
        self.camera_center = np.linalg.inv(self.world_view_transform.astype(np.float64))[3, :3].astype(np.float32)
        self.tanfovx = math.tan(self.FoVx * 0.5)
        self.tanfovy = math.tan(self.FoVy * 0.5)


def rot_xyz(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_camera(W, H, f, seed=None):
    """seed None: the SURVEY 8d camera (R=I, T=0, looks +z). Otherwise a small seeded
    perturbation (<= ~6 deg, <= 0.3 units) so that the view matrix is a general rigid motion."""
    if seed is None:
        return Camera(W, H, f, f)
    rng = np.random.default_rng(1000 + seed)
    ang = rng.uniform(-0.1, 0.1, 3)
    T = rng.uniform(-0.3, 0.3, 3)
    return Camera(W, H, f, f, rot_xyz(*ang), T)


def make_gaussians(P, cam, seed=0, degree_mode="all3", scale_mu=0.012, scale_sigma=0.6, behind_frac=0.02,
                   zmin=2.0, zmax=12.0):
    """SURVEY.md 8d Gaussian recipe (positions given in the R=I,T=0 camera frame == world frame).
    degree_mode: 'all3' | 'all0' | 'mixed' (categorical p=(.45,.2,.15,.2), tails of features_rest zeroed).
    Returns a dict of float32/int32 numpy arrays in the layout the rasterizer boundary expects:
    means3D[P,3], opacity[P,1] RAW, scales[P,3] ACTIVATED, rotations[P,4] UNIT, sh[P,16,3], degrees[P,1]."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(zmin, zmax, P)
    nb = int(round(P * behind_frac))
    if nb:
        idx = rng.choice(P, nb, replace=False)
        z[idx] = rng.uniform(-1.0, 0.2, nb)
    u, v = rng.uniform(-1, 1, P), rng.uniform(-1, 1, P)
    x = u * z * cam.tanfovx * 1.15
    y = v * z * cam.tanfovy * 1.15
    means = np.stack([x, y, z], 1).astype(np.float32)
    scales = np.exp(rng.normal(math.log(scale_mu), scale_sigma, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacity = rng.normal(0, 2.0, (P, 1)).astype(np.float32)
    sh = np.concatenate([rng.normal(0, 0.5, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3))], 1).astype(np.float32)
    if degree_mode == "all3":
        deg = np.full((P, 1), 3, np.int32)
    elif degree_mode == "all0":
        deg = np.zeros((P, 1), np.int32)
    elif degree_mode == "mixed":
        deg = rng.choice(4, size=(P, 1), p=[0.45, 0.2, 0.15, 0.2]).astype(np.int32)
    else:
        raise ValueError(degree_mode)
    K = (deg[:, 0] + 1) ** 2
    sh[np.arange(16)[None, :] >= K[:, None]] = 0.0
    return dict(means3D=means, opacity=opacity, scales=scales, rotations=q, sh=sh, degrees=deg)


def make_gaussians_clustered(P, cam, seed=0, degree_mode="mixed", fg_frac=0.62, fg_extent=0.39, fg_depth=(3.0, 0.45),
                             fg_scale_mu=0.005, bg_scale_mu=0.014, sky_frac=0.33, behind_frac=0.02):
    """A scene with the load distribution of a real capture (full_eval.py:21-24: garden, bicycle, train ... are an object
    in front of a large background under an empty sky), for the questions a uniform-in-frustum cloud cannot ask: tile
    lists many times the mean next to near-empty tiles.
      * fg_frac of the Gaussians: a dense foreground blob -- a 3D normal cloud whose 2-sigma footprint covers about
        fg_extent^2 = 15 % of the image (centre slightly below the middle), small log-normal splats;
      * 1 %: a thin "ground haze" of large, mostly transparent splats in front of everything (long lists of weak entries);
      * the rest: background, uniform over the image BELOW the sky line at depths 6..40, log-normal scales with a heavy
        large-splat tail (scale ~ depth, as a densified background has);
      * the top sky_frac of the image holds 0.3 % of the Gaussians only.
    Same layout of the returned dict as make_gaussians."""
    rng = np.random.default_rng(seed)
    n_fg = int(round(P * fg_frac))
    n_haze = int(round(P * 0.01))
    n_sky = int(round(P * 0.003))
    n_bg = P - n_fg - n_haze - n_sky
    tx, ty = cam.tanfovx, cam.tanfovy
    # foreground blob: normalised image coordinates u, v in [-1, 1]; 2 sigma = fg_extent
    zf = np.abs(rng.normal(fg_depth[0], fg_depth[1], n_fg)) + 0.5
    uf = rng.normal(0.0, fg_extent / 2, n_fg)
    vf = rng.normal(0.15, fg_extent / 2, n_fg)
    sf = np.exp(rng.normal(math.log(fg_scale_mu), 0.7, (n_fg, 3)))
    # haze: near, large, weak
    zh = rng.uniform(1.0, 2.5, n_haze)
    uh, vh = rng.uniform(-1, 1, n_haze), rng.uniform(-1 + 2 * sky_frac, 1, n_haze)
    sh_ = np.exp(rng.normal(math.log(0.025), 0.5, (n_haze, 3)))
    # background below the sky line
    zb = np.exp(rng.uniform(math.log(6.0), math.log(40.0), n_bg))
    ub, vb = rng.uniform(-1.1, 1.1, n_bg), rng.uniform(-1 + 2 * sky_frac, 1.1, n_bg)
    sb = np.exp(rng.normal(math.log(bg_scale_mu), 0.8, (n_bg, 3))) * (zb[:, None] / 10.0)
    # sky: a few far splats
    zs = rng.uniform(30.0, 80.0, n_sky)
    us, vs = rng.uniform(-1, 1, n_sky), rng.uniform(-1, -1 + 2 * sky_frac, n_sky)
    ssz = np.exp(rng.normal(math.log(0.2), 0.5, (n_sky, 3)))
    z = np.concatenate([zf, zh, zb, zs])
    u = np.concatenate([uf, uh, ub, us])
    v = np.concatenate([vf, vh, vb, vs])
    scales = np.concatenate([sf, sh_, sb, ssz]).astype(np.float32)
    opacity = rng.normal(0, 2.0, (P, 1))
    opacity[n_fg:n_fg + n_haze] = rng.normal(-3.0, 1.0, (n_haze, 1))   # sigmoid(-3) = 0.05: weak entries
    perm = rng.permutation(P)   # memory order is not depth or cluster order (a trained scene's is not either)
    z, u, v, scales, opacity = z[perm], u[perm], v[perm], scales[perm], opacity[perm]
    nb = int(round(P * behind_frac))
    if nb:
        z[rng.choice(P, nb, replace=False)] = rng.uniform(-1.0, 0.2, nb)
    means = np.stack([u * z * tx, v * z * ty, z], 1).astype(np.float32)
    q = rng.normal(0, 1, (P, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    sh = np.concatenate([rng.normal(0, 0.5, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3))], 1).astype(np.float32)
    if degree_mode == "all3":
        deg = np.full((P, 1), 3, np.int32)
    elif degree_mode == "all0":
        deg = np.zeros((P, 1), np.int32)
    elif degree_mode == "mixed":
        deg = rng.choice(4, size=(P, 1), p=[0.45, 0.2, 0.15, 0.2]).astype(np.int32)
    else:
        raise ValueError(degree_mode)
    K = (deg[:, 0] + 1) ** 2
    sh[np.arange(16)[None, :] >= K[:, None]] = 0.0
    return dict(means3D=means, opacity=opacity.astype(np.float32), scales=scales, rotations=q, sh=sh, degrees=deg)


def upstream_grad(W, H, seed=1):
    """Cheap seeded dL/d(out_color) ~ N(0,1)/N for kernel-level parity and timing."""
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, (3, H, W)) / (W * H)).astype(np.float32)


# named workloads: BASELINE.json configs / SURVEY 8d substitutions
WORKLOADS = {
    "cfg0_10k_400": dict(P=10_000, W=400, H=400, f=300.0, degree_mode="all0"),
    "lego_like_300k_800": dict(P=300_000, W=800, H=800, f=600.0, degree_mode="all3"),
    "metric_500k_1600x1062": dict(P=500_000, W=1600, H=1062, f=1200.0, degree_mode="all3"),
    "garden_like_2M_1600x1062": dict(P=2_000_000, W=1600, H=1062, f=1200.0, degree_mode="mixed"),
    # configs[3] / configs[4] stand-ins: smaller splats, as a densified scene has (R stays ~10 per Gaussian)
    "bicycle_like_5M_1600x1062": dict(P=5_000_000, W=1600, H=1062, f=1200.0, degree_mode="mixed", scale_mu=0.008),
    "train_like_6M_1920x1080": dict(P=6_000_000, W=1920, H=1080, f=1400.0, degree_mode="mixed", scale_mu=0.008),
    # real-scene-shaped load: a dense foreground object on ~15 % of the image, a large-splat background, an
    # empty sky -- tile lists an order of magnitude above the mean next to near-empty tiles (make_gaussians_clustered)
    "clustered_500k_1600x1062": dict(P=500_000, W=1600, H=1062, f=1200.0, degree_mode="all3", clustered=True),
    "garden_clustered_2M": dict(P=2_000_000, W=1600, H=1062, f=1200.0, degree_mode="mixed", clustered=True),
}


def make_workload(name, seed=0):
    """-> (workload dict, camera 0, gaussians) of a named workload."""
    w = WORKLOADS[name]
    cam = make_camera(w["W"], w["H"], w["f"], None)
    if w.get("clustered"):
        g = make_gaussians_clustered(w["P"], cam, seed=seed, degree_mode=w["degree_mode"])
    else:
        g = make_gaussians(w["P"], cam, seed=seed, degree_mode=w["degree_mode"], scale_mu=w.get("scale_mu", 0.012))
    return w, cam, g
