"""CPU execution of the PRODUCT's __host__ __device__ math (reduced-3dgs_amd/csrc/gauss_math.h, blend_math.h)
through the test shim tests/hostcheck/hostcheck.hip, against the oracle.  No GPU needed: this catches
transcription errors in the per-lane arithmetic before any GPU time is spent.  The kernels' cooperative
parts (LDS staging, DPP reductions, atomics, sorts) are covered by the -m gpu tests only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import synth_scene as ss
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "hostcheck.hip")
SO = os.path.join(HERE, "hostcheck", "libhostcheck.so")
HIPCC = "/opt/rocm/bin/hipcc"


def _lib():
    hdrs = [os.path.join(HERE, "..", "reduced-3dgs_amd", "csrc", h) for h in ("gauss_math.h", "blend_math.h", "common.h")]
    newest = max(os.path.getmtime(p) for p in [SRC] + hdrs)
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available to build the host-check shim")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-o", SO, SRC])
    return C.CDLL(SO)


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


CASES = [
    dict(P=500, W=72, H=50, f=60.0, cam_seed=None, gseed=0, degree_mode="all3", lam=0.0, spread=1.0),
    dict(P=500, W=70, H=45, f=60.0, cam_seed=2, gseed=1, degree_mode="mixed", lam=0.1, spread=1.0),
    dict(P=400, W=64, H=64, f=50.0, cam_seed=4, gseed=2, degree_mode="all0", lam=0.0, spread=1.35),
    dict(P=500, W=70, H=45, f=60.0, cam_seed=3, gseed=5, degree_mode="mixed", lam=0.0, spread=1.0),
]


@pytest.mark.parametrize("kw", CASES, ids=["deg3", "mixed_sparsity", "deg0_clamp", "mixed"])
@pytest.mark.parametrize("precomp", [False, True], ids=["sh_scale_rot", "precomp_colour_cov"])
def test_device_math_on_host_matches_oracle(kw, precomp):
    L = _lib()
    W, H, P = kw["W"], kw["H"], kw["P"]
    cam = ss.make_camera(W, H, kw["f"], kw["cam_seed"])
    g = ss.make_gaussians(P, cam, seed=kw["gseed"], degree_mode=kw["degree_mode"], scale_mu=0.12, scale_sigma=0.7)
    g["means3D"][:, :2] *= kw["spread"]
    bg = np.array([0.3, 0.5, 0.7], np.float32)
    rng = np.random.default_rng(5)
    colors = cov = None
    sh, scales, rots = g["sh"], g["scales"], g["rotations"]
    if precomp:
        colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
        o0 = orc.forward(bg, g["means3D"], None, g["opacity"], scales, rots, 1.0, None, cam.world_view_transform,
                         cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, sh, g["degrees"], cam.camera_center)
        cov = o0["state"]["cov3D"].copy()
        cov[o0["radii"] == 0] = np.array([1e-2, 0, 0, 1e-2, 0, 1e-2], np.float32)
        sh = scales = rots = None
    out = orc.forward(bg, g["means3D"], colors, g["opacity"], scales, rots, 1.0, cov, cam.world_view_transform,
                      cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, sh, g["degrees"], cam.camera_center,
                      want_ambig=True)
    st = out["state"]
    M = 0 if sh is None else 16
    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), np.float32)
    depths = np.zeros(P, np.float32)
    conic_op = np.zeros((P, 4), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    cbits = np.zeros(P, np.uint32)
    tiles = np.zeros(P, np.uint32)
    rect = np.zeros((P, 4), np.int32)
    deg = np.ascontiguousarray(g["degrees"].reshape(-1))
    view, proj, campos = (np.ascontiguousarray(a, np.float32) for a in
                          (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
    opac = np.ascontiguousarray(g["opacity"].reshape(-1))
    L.hc_preprocess(C.c_int(P), C.c_int(M), p(deg), p(g["means3D"]), p(scales), C.c_float(1.0), p(rots), p(opac),
                    p(sh), p(cov), p(colors), p(view), p(proj), p(campos), C.c_int(W), C.c_int(H),
                    C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(radii), p(xy), p(depths), p(conic_op), p(rgb),
                    p(cbits), p(tiles), p(rect))
    vis = out["radii"] > 0
    assert vis.sum() > 50
    # integer outputs + everything computed without exp(): bit-exact
    np.testing.assert_array_equal(radii, out["radii"])
    np.testing.assert_array_equal(tiles, st["tiles_touched"])
    np.testing.assert_array_equal(xy[vis], st["xy"][vis])
    np.testing.assert_array_equal(depths[vis], st["depths"][vis])
    np.testing.assert_array_equal(conic_op[vis, :3], st["conic_op"][vis, :3])
    np.testing.assert_allclose(conic_op[vis, 3], st["conic_op"][vis, 3], rtol=3e-7)  # expf implementations
    feat = colors if precomp else st["rgb"]
    np.testing.assert_array_equal(rgb[vis], feat[vis])
    if not precomp:
        bits = st["clamped"][:, 0] | (st["clamped"][:, 1] << 1) | (st["clamped"][:, 2] << 2)
        np.testing.assert_array_equal(cbits[vis], bits[vis])
    area = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    np.testing.assert_array_equal(area[vis], st["tiles_touched"][vis])

    # ---- blend forward on the oracle's binning -------------------------------------------------
    N = W * H
    color = np.zeros((3, H, W), np.float32)
    final_T = np.zeros(N, np.float32)
    n_contrib = np.zeros(N, np.uint32)
    L.hc_blend_fwd(C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(feat),
                   p(st["conic_op"]), p(bg), p(color), p(final_T), p(n_contrib), C.c_int(0), None, None)
    # the kernels' region pre-test (blend_math.h region_may_contribute) must not change a single bit
    color2, final_T2, n_contrib2 = np.zeros_like(color), np.zeros_like(final_T), np.zeros_like(n_contrib)
    skipped, total = C.c_long(0), C.c_long(0)
    L.hc_blend_fwd(C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(st["xy"]), p(feat),
                   p(st["conic_op"]), p(bg), p(color2), p(final_T2), p(n_contrib2), C.c_int(1), C.byref(skipped),
                   C.byref(total))
    np.testing.assert_array_equal(color2, color)
    np.testing.assert_array_equal(final_T2, final_T)
    np.testing.assert_array_equal(n_contrib2, n_contrib)
    assert skipped.value > 0.2 * total.value  # and it must actually remove work
    ok = out["ambig"].reshape(-1) == 0
    assert ok.mean() > 0.99
    assert np.abs(color.reshape(3, -1) - out["color"].reshape(3, -1))[:, ok].max() < 1e-5
    np.testing.assert_array_equal(n_contrib[ok], st["n_contrib"][ok])

    # ---- blend backward -----------------------------------------------------------------------
    dl = ss.upstream_grad(W, H, seed=9) * N
    gr = orc.backward(st, dl, kw["lam"])
    acc = np.zeros((P, 9), np.float64)
    L.hc_blend_bwd(C.c_int(P), C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(bg), p(st["xy"]),
                   p(st["conic_op"]), p(feat), p(st["final_T"]), p(st["n_contrib"]), p(dl), p(acc), C.c_int(0))
    acc2 = np.zeros((P, 9), np.float64)
    L.hc_blend_bwd(C.c_int(P), C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(bg), p(st["xy"]),
                   p(st["conic_op"]), p(feat), p(st["final_T"]), p(st["n_contrib"]), p(dl), p(acc2), C.c_int(1))
    np.testing.assert_array_equal(acc2, acc)

    def close(name, ref, got, rel=1e-4):
        scale = np.abs(ref).max() + 1e-30
        err = np.abs(ref - got).max()
        assert err <= rel * scale, f"{name}: {err:.3e} vs {scale:.3e}"

    close("dmean2D", gr["dL_dmeans2D"][:, :2], acc[:, :2])
    close("dconic", gr["dL_dconic"][:, [0, 1, 3]], acc[:, 2:5])
    close("dcolor", gr["dL_dcolors"], acc[:, 6:9])

    # ---- per-Gaussian backward fed with the oracle's 2D-stage gradients -------------------------
    d3 = np.zeros((P, 3), np.float32)
    dcov = np.zeros((P, 6), np.float32)
    dsh = np.zeros((P, max(M, 1), 3), np.float32)
    dsc = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32)
    dop = acc[:, 5].astype(np.float32).copy()  # pre-sigmoid-chain opacity gradient = blend-stage sum
    dm2 = np.ascontiguousarray(gr["dL_dmeans2D"])
    dcon = np.ascontiguousarray(gr["dL_dconic"])
    dcol = np.ascontiguousarray(gr["dL_dcolors"])
    L.hc_preprocess_bwd(C.c_int(P), C.c_int(M), p(deg), p(g["means3D"]), p(radii), p(sh), p(cbits), p(scales),
                        p(rots), C.c_float(1.0), p(cov), p(view), p(proj), p(campos), C.c_int(W), C.c_int(H),
                        C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(dm2), p(st["conic_op"]), p(dcon), p(dcol),
                        C.c_float(kw["lam"]), p(d3), p(dcov), p(dsh), p(dsc), p(drot), p(dop))
    # without a sparsity term the product takes the SH direction derivatives from the forward: same bits as the direct form
    L.hc_cached_sh_mismatches.restype = C.c_long
    assert L.hc_cached_sh_mismatches() == 0
    close("dmean3D", gr["dL_dmeans3D"], d3, 1e-5)
    close("dcov3D", gr["dL_dcov3D"], dcov, 1e-5)
    close("dopacity", gr["dL_dopacity"].reshape(-1), dop)
    if not precomp:
        close("dsh", gr["dL_dsh"], dsh, 1e-5)
        close("dscale", gr["dL_dscales"], dsc, 1e-5)
        close("drot", gr["dL_drotations"], drot, 1e-5)

    # ---- the product's default: the covariance chain in DOUBLE (gauss_math.h cov2d_backward_f64 / cov3d_backward_f64),
    # against oracle/backward_f64.c's independently derived chain fed the SAME fp32 2D-stage gradients: what is left is the
    # one rounding of each output to fp32 (and of the covariance part of dL/dmean before the fp32 projection / SH parts
    # are added to it)
    acc_in = np.zeros((P, 9), np.float64)
    acc_in[:, 0:2] = dm2[:, :2]
    acc_in[:, 2:5] = dcon[:, [0, 1, 3]]
    acc_in[:, 5] = acc[:, 5].astype(np.float32)
    acc_in[:, 6:9] = dcol
    want = orc.preprocess_bwd_f64(st, acc_in, kw["lam"])
    d3b, dcovb, dshb = np.zeros_like(d3), np.zeros_like(dcov), np.zeros_like(dsh)
    dscb, drotb = np.zeros_like(dsc), np.zeros_like(drot)
    dopb = acc[:, 5].astype(np.float32).copy()
    L.hc_set_f64_chain(C.c_int(1))
    try:
        L.hc_preprocess_bwd(C.c_int(P), C.c_int(M), p(deg), p(g["means3D"]), p(radii), p(sh), p(cbits), p(scales),
                            p(rots), C.c_float(1.0), p(cov), p(view), p(proj), p(campos), C.c_int(W), C.c_int(H),
                            C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(dm2), p(st["conic_op"]), p(dcon), p(dcol),
                            C.c_float(kw["lam"]), p(d3b), p(dcovb), p(dshb), p(dscb), p(drotb), p(dopb))
    finally:
        L.hc_set_f64_chain(C.c_int(0))
    close("f64 chain dcov3D", want["dL_dcov3D"], dcovb, 2e-7)
    close("f64 chain dmean3D", want["dL_dmeans3D"], d3b, 1e-6)
    if not precomp:
        close("f64 chain dscale", want["dL_dscales"], dscb, 2e-7)
        close("f64 chain drot", want["dL_drotations"], drotb, 2e-7)
        np.testing.assert_array_equal(dshb, dsh)      # the SH part is untouched by the switch


def test_region_pretest_is_conservative_fuzz():
    """2M random splats (eigenvalue ratios up to 1e5, opacities down to 1e-4) x one 8x8 block: the kernels'
    skip test must never fire when any pixel of the block would blend, and should be tight otherwise."""
    L = _lib()
    L.hc_region_fuzz.restype = C.c_long
    skip, keep_empty = C.c_long(0), C.c_long(0)
    n = 2_000_000
    bad = L.hc_region_fuzz(C.c_long(n), C.c_uint(1234), C.byref(skip), C.byref(keep_empty))
    assert bad == 0
    assert skip.value > 0.3 * n            # the fuzz distribution exercises the skip path
    assert keep_empty.value < 0.02 * n     # exact minimisation: almost no false keeps


@pytest.mark.parametrize("kw", [
    dict(P=4000, W=200, H=136, f=150.0, cam_seed=3, gseed=5, scale_mu=0.05, op_mu=0.0, gain=0.85),
    dict(P=3000, W=160, H=120, f=120.0, cam_seed=None, gseed=6, scale_mu=0.15, op_mu=-3.0, gain=0.7),   # faint, large splats
    dict(P=3000, W=176, H=96, f=130.0, cam_seed=8, gseed=7, scale_mu=0.02, op_mu=4.0, gain=0.95),        # opaque, small splats
], ids=["mid", "faint_large", "opaque_small"])
def test_opacity_aware_rects_change_no_pixel_decision(kw):
    """The product bins a Gaussian into the part of the reference's 3-sigma square that it can reach with alpha >= 1/255
    (gauss_math.h tighten_rect, run here on the CPU).  Claim: the shorter lists give the reference's results.  Checked
    with the oracle alone: (1) every rect lies inside the reference's, (2) no pixel of a left-out tile passes the
    reference's per-pixel test (exhaustive), (3) the oracle's blend over the shorter lists returns the image and final T
    of the full lists BIT FOR BIT and the same gradients, (4) the lists really are shorter."""
    L = _lib()
    W, H, P = kw["W"], kw["H"], kw["P"]
    cam = ss.make_camera(W, H, kw["f"], kw["cam_seed"])
    g = ss.make_gaussians(P, cam, seed=kw["gseed"], degree_mode="mixed", scale_mu=kw["scale_mu"], scale_sigma=0.8)
    g["opacity"] = (g["opacity"] + kw["op_mu"]).astype(np.float32)
    # a few extreme aspect ratios and opacities right at the 1/255 threshold
    g["scales"][:50, 0] *= 30.0
    g["opacity"][50:80] = np.float32(np.log((1 / 255.0) / (1 - 1 / 255.0)))
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    fwd = lambda rects=None: orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                                         cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W,
                                         g["sh"], g["degrees"], cam.camera_center, rects=rects)
    ref = fwd()
    radii = np.zeros(P, np.int32)
    rects = np.zeros((P, 4), np.uint16)
    tiles = np.zeros(P, np.uint32)
    tiles_ref = np.zeros(P, np.uint32)
    view, proj, campos = (np.ascontiguousarray(a, np.float32) for a in
                          (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
    L.hc_tight_rects(C.c_int(P), p(g["means3D"]), p(g["scales"]), C.c_float(1.0), p(g["rotations"]),
                     p(np.ascontiguousarray(g["opacity"].reshape(-1))), p(view), p(proj), p(campos), C.c_int(W), C.c_int(H),
                     C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(radii), p(rects), p(tiles), p(tiles_ref))
    np.testing.assert_array_equal(radii, ref["radii"])                      # the radii output is the reference's
    np.testing.assert_array_equal(tiles_ref, ref["state"]["tiles_touched"])  # and so is what num_rendered counts
    bad, left = orc.culled_tile_violations(ref["state"], rects)
    assert bad == 0, f"{bad} pixels of left-out tiles would have been blended by the reference"
    tight = fwd(rects)
    assert tight["state"]["pairs"] == int(tiles[radii > 0].sum()) == ref["num_rendered"] - left
    assert tight["state"]["pairs"] < kw["gain"] * ref["num_rendered"], "the rects are not tighter than the reference's"
    np.testing.assert_array_equal(tight["color"], ref["color"])
    np.testing.assert_array_equal(tight["state"]["final_T"], ref["state"]["final_T"])
    dl = ss.upstream_grad(W, H, seed=3) * (W * H)
    ga, gb = orc.backward(ref["state"], dl, 0.05), orc.backward(tight["state"], dl, 0.05)
    for k in ga:
        np.testing.assert_array_equal(ga[k], gb[k], err_msg=k)


@pytest.mark.parametrize("kw", [
    dict(P=20_000, W=640, H=360, f=800.0, seed=3),
    dict(P=20_000, W=1280, H=720, f=900.0, seed=5),   # one pixel of a left-out tile passed the fp32 test before the margin
], ids=["needles_640x360", "needles_1280x720"])
def test_opacity_aware_rects_needle_scene(kw):
    """20k needles (one scale 0.3..3, the other two 1e-4; median radius ~500 px).  For such a splat seen at an
    angle the reference's fp32 evaluation of the conic form is a small difference of large products: a pixel whose exact
    alpha is 20 % below 1/255 can pass the test.  The rect's margin scales with the conic's conditioning (gauss_math.h
    kCancelMargin): no pixel of any left-out tile may pass the reference's (the oracle's fp32) per-pixel test, and the
    rects must still remove a large part of the reference's pairs."""
    L = _lib()
    W, H, P = kw["W"], kw["H"], kw["P"]
    cam = ss.make_camera(W, H, kw["f"], 3)
    g = ss.make_gaussians(P, cam, seed=kw["seed"], degree_mode="all0", scale_mu=0.02)
    rng = np.random.default_rng(kw["seed"])
    g["scales"][:, 0] = rng.uniform(0.3, 3.0, P).astype(np.float32)
    g["scales"][:, 1:] = 1e-4
    bg = np.zeros(3, np.float32)
    ref = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center)
    radii = np.zeros(P, np.int32)
    rects = np.zeros((P, 4), np.uint16)
    tiles = np.zeros(P, np.uint32)
    tiles_ref = np.zeros(P, np.uint32)
    view, proj, campos = (np.ascontiguousarray(a, np.float32) for a in
                          (cam.world_view_transform, cam.full_proj_transform, cam.camera_center))
    L.hc_tight_rects(C.c_int(P), p(g["means3D"]), p(g["scales"]), C.c_float(1.0), p(g["rotations"]),
                     p(np.ascontiguousarray(g["opacity"].reshape(-1))), p(view), p(proj), p(campos), C.c_int(W), C.c_int(H),
                     C.c_float(cam.tanfovx), C.c_float(cam.tanfovy), p(radii), p(rects), p(tiles), p(tiles_ref))
    np.testing.assert_array_equal(radii, ref["radii"])
    assert np.median(radii[radii > 0]) > 400
    bad, left = orc.culled_tile_violations(ref["state"], rects)
    assert bad == 0, f"{bad} pixels of left-out tiles would have been blended by the reference"
    assert left > 0.4 * ref["num_rendered"]


def test_unit_lists_cover_every_tile_once():
    """The backward blend's eight unit lists (common.h TileGrid: list g = the 4 x 4 tile blocks g, g + 8, ... of the image):
    together every tile exactly once, none longer than list_tiles_max, the per-list fit of a pass below the per-list capacity
    of any reservation that holds its pairs -- for grids with ragged edges, strips, and fewer blocks than lists."""
    lib = _lib()
    lib.hc_unit_list.restype = C.c_int
    rng = np.random.default_rng(0)
    shapes = [(1, 1), (1, 9), (9, 1), (3, 3), (4, 4), (5, 5), (8, 8), (100, 67), (120, 68), (129, 69), (7, 300), (1024, 1024)]
    shapes += [tuple(int(v) for v in rng.integers(1, 200, 2)) for _ in range(20)]
    for gx, gy in shapes:
        Tn = gx * gy
        seen = np.zeros(Tn, np.int64)
        info = np.zeros(4, np.int32)
        pairs = int(rng.integers(0, 40 * Tn + 1))
        reserve = pairs + int(rng.integers(0, 3 * Tn + 1))
        longest = 0
        for g_ in range(8):
            tiles = np.full(Tn + 16 * 8, -2, np.int32)
            n = lib.hc_unit_list(gx, gy, g_, p(tiles), len(tiles), C.c_uint(pairs), C.c_uint(max(reserve, 1)), p(info))
            t = tiles[:n]
            assert n % 16 == 0 and np.all(t >= -1) and np.all(t < Tn)
            valid = t[t >= 0]
            np.add.at(seen, valid, 1)
            longest = max(longest, len(valid))
            # a block is 4 x 4 neighbouring tiles: the 16 slots of a block lie in one 4-aligned square
            for b in range(n // 16):
                v = t[16 * b:16 * b + 16]
                v = v[v >= 0]
                if len(v):
                    assert len(set((v % gx) // 4)) == 1 and len(set((v // gx) // 4)) == 1
                    k = (v[0] // gx // 4) * ((gx + 3) // 4) + (v[0] % gx) // 4
                    assert k % 8 == g_ and k // 8 == b
        assert np.all(seen == 1), (gx, gy)
        slots, tiles_max, fit, cap = (int(v) for v in info)
        assert longest <= tiles_max <= Tn
        assert fit == tiles_max + min(pairs >> 7, 8 * Tn) // 8
        assert cap % 8 == 0 and fit < cap // 8    # a list's units fit its slots whatever segment length it has to take


@pytest.mark.parametrize("S,max_seg", [(128, 32), (256, 32), (128, 2)], ids=["S128", "S256", "S128_capped_at_2"])
def test_backward_in_list_segments_from_forward_checkpoints(S, max_seg):
    """The list split of the backward blend on the CPU (blend.hip: tiles with long lists are walked in segments of S entries
    by several workgroups, each from the (T, accumulated colour) checkpoint the forward left at its end): the kernels' own
    per-lane functions and the kernel's own start-from-checkpoint arithmetic, pixel by pixel, against the whole-list walk of
    the same functions.  Every entry belongs to one segment, so the per-Gaussian sums must agree to rounding (the state at
    a segment's end is the forward's running product and a colour difference instead of the backward's division chain);
    a scene with lists several segments deep, the segment cap (the last segment takes the rest), and both against the
    oracle's backward at the parity bar."""
    L = _lib()
    W, H, P = 64, 48, 8000
    cam = ss.make_camera(W, H, 55.0, 3)
    g = ss.make_gaussians(P, cam, seed=12, degree_mode="all0", scale_mu=0.03, zmin=2.0, zmax=6.0)
    g["opacity"] -= 2.5    # weak entries: the pixels stay live deep into the lists
    bg = np.array([0.2, 0.3, 0.1], np.float32)
    out = orc.forward(bg, g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None, cam.world_view_transform,
                      cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"], g["degrees"], cam.camera_center,
                      want_ambig=True)
    st = out["state"]
    assert st["n_contrib"].max() > 3 * S or max_seg == 2, "the scene must have lists several segments deep"
    N = W * H
    amb = out["ambig"].reshape(-1) != 0
    dl = (ss.upstream_grad(W, H, seed=9) * N).reshape(3, -1)
    dl[:, amb] = 0.0    # (threshold-ambiguous pixels take no part in a comparison with the oracle)
    dl = np.ascontiguousarray(dl.reshape(3, H, W))
    args = (C.c_int(P), C.c_int(W), C.c_int(H), p(st["ranges"]), p(st["point_list"]), p(bg), p(st["xy"]),
            p(st["conic_op"]), p(st["rgb"]), p(st["final_T"]), p(st["n_contrib"]), p(dl))
    whole = np.zeros((P, 9), np.float64)
    L.hc_blend_bwd(*args, p(whole), C.c_int(0))
    split = np.zeros((P, 9), np.float64)
    n_split = C.c_long(0)
    L.hc_blend_bwd_segments(*args, p(split), C.c_int(S), C.c_int(max_seg), C.byref(n_split))
    assert n_split.value > 0.3 * N, "most pixels must pass through at least one segment end"
    names = ["dmean2D.x", "dmean2D.y", "dconic.a", "dconic.b", "dconic.c", "dopacity", "dcolor.r", "dcolor.g", "dcolor.b"]
    worst = 0.0
    for k, n in enumerate(names):
        scale = np.abs(whole[:, k]).max() + 1e-30
        err = np.abs(split[:, k] - whole[:, k]).max() / scale
        worst = max(worst, err)
        assert err <= 2e-5, f"{n}: segments vs whole list {err:.2e} of the tensor's maximum"
    assert worst > 0.0   # (it IS a different rounding: an exact match would mean nothing was started from a checkpoint)
    gr = orc.backward(st, dl, 0.0)
    for name, ref, got in (("dmean2D", gr["dL_dmeans2D"][:, :2], split[:, :2]), ("dconic", gr["dL_dconic"][:, [0, 1, 3]], split[:, 2:5]),
                           ("dcolor", gr["dL_dcolors"], split[:, 6:9])):
        err = np.abs(ref - got).max() / (np.abs(ref).max() + 1e-30)
        assert err <= 1e-4, f"{name}: segment walk vs the oracle {err:.2e}"
    # one segment per tile IS the whole-list walk (the double accumulators add the pixels tile by tile here, row by row there)
    one = np.zeros((P, 9), np.float64)
    L.hc_blend_bwd_segments(*args, p(one), C.c_int(1 << 20), C.c_int(max_seg), C.byref(n_split))
    assert n_split.value == 0
    np.testing.assert_allclose(one, whole, rtol=1e-12, atol=1e-12 * np.abs(whole).max())
