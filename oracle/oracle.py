"""ctypes/numpy front-end of the CPU oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  Parity status:
"parity unpinned" against the reference CUDA kernels (see raster_oracle.c header).

The two entry points mirror the reference's torch-C++ operators
(DGR/rasterize_points.cu:136-222 RasterizeGaussiansCUDA, :224-305
RasterizeGaussiansBackwardCUDA, :43-134 the ragged-SH inference variant) on numpy
arrays, so parity tests read like calls into the reference `_C` module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# R3_ORACLE_VARIANT=fma: the restatement built with FMA contraction allowed (Makefile; tools/contraction_flips.py runs the
# two builds in two processes).  Every test uses the default: one rounding per operation.
_VARIANT = os.environ.get("R3_ORACLE_VARIANT", "")
_SO = "libraster_oracle.so" if not _VARIANT else f"libraster_oracle_{_VARIANT}.so"


def build(force=False):
    so = os.path.join(_HERE, _SO)
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "backward_f64.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", _SO], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_bin.restype = C.c_int64
        _LIB.orc_bin_rects.restype = C.c_int64
        _LIB.orc_culled_tile_violations.restype = C.c_int64
        _LIB.orc_higher_msb.restype = C.c_uint32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if a.size else None


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


TILE = 16


def forward(bg, means3D, colors_precomp, opacity, scales, rotations, scale_modifier, cov3D_precomp,
            viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degrees, campos,
            ragged=None, counter_mode=False, want_ambig=False, ambig_rel=1e-4, rects=None, geometry_only=False):
    """Oracle of `_C.rasterize_gaussians` (and `_variableSH_bands` when `ragged`
    = (coeffs_num, per_band_count, cumsum_count) and `sh` is the flat ragged buffer).
    Absent optional inputs: None or empty arrays.  Returns a dict with the public
    outputs (num_rendered, color[3,H,W], radii[P]) and the internal state needed by
    `backward` / bit-exact binning checks (keys, point_list, ranges, n_contrib, final_T...).
    rects ([P,4] tile rects x0, y0, x1, y1 with exclusive maxima): bin every visible Gaussian into THAT rect instead of
    the reference's 3-sigma square (see orc_bin_rects); `culled_tile_violations` says whether the rects are admissible.
    num_rendered stays the reference's count, state["pairs"] is the length of the lists.
    geometry_only: stop after the per-Gaussian stage and the pair COUNT (radii, tiles_touched, depths, conics, colours and
    num_rendered are valid; no lists, no image) -- what the 5 M / 6 M property tests compare element-wise."""
    L = lib()
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    bg = _f32(bg)
    colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (colors_precomp, scales, rotations, cov3D_precomp))
    sh = _f32(sh)
    opacity = _f32(opacity)
    vm, pm, campos = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    degrees = _i32(degrees) if degrees is not None else np.zeros(P, np.int32)
    M = 0
    coeffs = perband = cumsum = None
    if ragged is not None:
        coeffs, perband, cumsum = map(_i32, ragged)
    elif sh is not None:
        M = sh.shape[1]
    N = W * H
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    st = dict(P=P, M=M, W=W, H=H, tan_fovx=float(tan_fovx), tan_fovy=float(tan_fovy), mod=float(scale_modifier),
              bg=bg, means3D=means3D, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
              cov3D_precomp=cov3D_precomp, vm=vm, pm=pm, campos=campos, sh=sh, degrees=degrees)
    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), np.float32)
    depths = np.zeros(P, np.float32)
    cov3D = np.zeros((P, 6), np.float32)
    conic_op = np.zeros((P, 4), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    clamped = np.zeros((P, 3), np.uint8)
    tiles = np.zeros(P, np.uint32)
    color = np.zeros((3, H, W), np.float32)
    final_T = np.zeros(N, np.float32)
    n_contrib = np.zeros(N, np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    R = pairs = 0
    keys = np.zeros(0, np.uint64)
    plist = np.zeros(0, np.uint32)
    touched = transm = ambig = None
    if P:
        L.orc_preprocess(C.c_int(P), C.c_int(M), _p(degrees), _p(means3D), _p(scales), C.c_float(scale_modifier),
                         _p(rotations), _p(opacity), _p(sh), _p(cov3D_precomp), _p(colors_precomp), _p(vm), _p(pm),
                         _p(campos), C.c_int(W), C.c_int(H), C.c_float(tan_fovx), C.c_float(tan_fovy),
                         _p(coeffs), _p(perband), _p(cumsum), _p(radii), _p(xy), _p(depths), _p(cov3D),
                         _p(conic_op), _p(rgb), _p(clamped), _p(tiles))
        R = int(L.orc_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(radii), _p(xy), _p(depths), _p(tiles),
                          None, None, None))
    if P and geometry_only:
        pairs = R
    elif P:
        if rects is not None:
            rects = np.ascontiguousarray(rects, dtype=np.uint16).reshape(P, 4)
        pairs = int(L.orc_bin_rects(C.c_int(P), C.c_int(W), C.c_int(H), _p(radii), _p(xy), _p(depths), _p(tiles),
                                    _p(rects), None, None, None))
        keys = np.zeros(max(pairs, 1), np.uint64)
        plist = np.zeros(max(pairs, 1), np.uint32)
        L.orc_bin_rects(C.c_int(P), C.c_int(W), C.c_int(H), _p(radii), _p(xy), _p(depths), _p(tiles), _p(rects),
                        _p(keys), _p(plist), _p(ranges))
        keys, plist = keys[:pairs], plist[:pairs]
        feat = colors_precomp if colors_precomp is not None else rgb
        if counter_mode:
            touched = np.zeros(P, np.int32)
            transm = np.zeros(P, np.float32)
        if want_ambig:
            ambig = np.zeros(N, np.uint8)
        L.orc_blend_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(plist), _p(xy), _p(feat), _p(conic_op), _p(bg),
                        _p(color), _p(final_T), _p(n_contrib), _p(touched), _p(transm), _p(ambig),
                        C.c_float(ambig_rel))
    else:
        color[:] = 0  # reference returns the zero-initialised image when P == 0 (rasterize_points.cu:170,185)
    st.update(radii=radii, xy=xy, depths=depths, cov3D=cov3D, conic_op=conic_op, rgb=rgb, clamped=clamped,
              tiles_touched=tiles, keys=keys, point_list=plist, ranges=ranges, final_T=final_T,
              n_contrib=n_contrib, num_rendered=R, pairs=pairs, rects=rects)
    out = dict(num_rendered=R, color=color, radii=radii, state=st)
    if counter_mode:
        out.update(touched_pixels=touched, transmittance=transm)
    if want_ambig:
        out["ambig"] = ambig.reshape(H, W)
    return out


def with_rects(out, rects, want_ambig=False, ambig_rel=1e-4):
    """The forward result `out` (of `forward`) re-binned into `rects` and re-blended: same per-Gaussian quantities, the
    reference's binning algorithm over the given rects, the reference's blend over those lists."""
    L = lib()
    st = dict(out["state"])
    P, W, H = st["P"], st["W"], st["H"]
    rects = np.ascontiguousarray(rects, dtype=np.uint16).reshape(P, 4)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    args = (C.c_int(P), C.c_int(W), C.c_int(H), _p(st["radii"]), _p(st["xy"]), _p(st["depths"]), _p(st["tiles_touched"]),
            _p(rects))
    pairs = int(L.orc_bin_rects(*args, None, None, None))
    keys, plist = np.zeros(max(pairs, 1), np.uint64), np.zeros(max(pairs, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.orc_bin_rects(*args, _p(keys), _p(plist), _p(ranges))
    color = np.zeros((3, H, W), np.float32)
    final_T, n_contrib = np.zeros(W * H, np.float32), np.zeros(W * H, np.uint32)
    ambig = np.zeros(W * H, np.uint8) if want_ambig else None
    feat = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
    L.orc_blend_fwd(C.c_int(W), C.c_int(H), _p(ranges), _p(plist), _p(st["xy"]), _p(feat), _p(st["conic_op"]),
                    _p(st["bg"]), _p(color), _p(final_T), _p(n_contrib), None, None, _p(ambig), C.c_float(ambig_rel))
    vis = st["radii"] > 0
    tiles = np.where(vis, (rects[:, 2].astype(np.int64) - rects[:, 0]) * (rects[:, 3].astype(np.int64) - rects[:, 1]), 0)
    st.update(keys=keys[:pairs], point_list=plist[:pairs], ranges=ranges, final_T=final_T, n_contrib=n_contrib,
              pairs=pairs, rects=rects, tiles_binned=tiles.astype(np.uint32))
    res = dict(num_rendered=out["num_rendered"], color=color, radii=out["radii"], state=st)
    if want_ambig:
        res["ambig"] = ambig.reshape(H, W)
    return res


def culled_tile_violations(st, rects):
    """(violations, tiles_left_out) of binning the state's Gaussians into `rects` ([P,4] tiles, exclusive maxima) instead
    of the reference's squares: violations counts the pixels of left-out tiles that the reference's per-pixel test would
    have blended (and rects sticking out of the reference's).  0 <=> the shorter lists take the same pixel decisions."""
    L = lib()
    rects = np.ascontiguousarray(rects, dtype=np.uint16).reshape(st["P"], 4)
    left = C.c_int64(0)
    bad = int(L.orc_culled_tile_violations(C.c_int(st["P"]), C.c_int(st["W"]), C.c_int(st["H"]), _p(st["radii"]),
                                           _p(st["xy"]), _p(st["conic_op"]), _p(rects), C.byref(left)))
    return bad, int(left.value)


def backward(st, dL_dout_color, lambda_sh_sparsity=0.0):
    """Oracle of `_C.rasterize_gaussians_backward`: returns the 8 gradient arrays in the
    reference order (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
    dL_dscales, dL_drotations) as a dict."""
    L = lib()
    P, M, W, H = st["P"], st["M"], st["W"], st["H"]
    g = _f32(dL_dout_color)
    dmean2D = np.zeros((P, 3), np.float32)
    dconic = np.zeros((P, 4), np.float32)
    dopac = np.zeros((P, 1), np.float32)
    dcolor = np.zeros((P, 3), np.float32)
    dmean3D = np.zeros((P, 3), np.float32)
    dcov3D = np.zeros((P, 6), np.float32)
    dsh = np.zeros((P, M, 3), np.float32)
    dscale = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32)
    if P:
        feat = st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]
        L.orc_blend_bwd(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["bg"]),
                        _p(st["xy"]), _p(st["conic_op"]), _p(feat), _p(st["final_T"]), _p(st["n_contrib"]),
                        _p(g), _p(dmean2D), _p(dconic), _p(dopac), _p(dcolor))
        cov = st["cov3D_precomp"] if st["cov3D_precomp"] is not None else st["cov3D"]
        L.orc_preprocess_bwd(C.c_int(P), C.c_int(M), _p(st["degrees"]), _p(st["means3D"]), _p(st["radii"]),
                             _p(st["sh"]), _p(st["clamped"]), _p(st["scales"]), _p(st["rotations"]),
                             C.c_float(st["mod"]), _p(cov), _p(st["vm"]), _p(st["pm"]), _p(st["campos"]),
                             C.c_int(W), C.c_int(H), C.c_float(st["tan_fovx"]), C.c_float(st["tan_fovy"]),
                             _p(dmean2D), _p(st["conic_op"]), _p(dconic), _p(dmean3D), _p(dcolor), _p(dcov3D),
                             _p(dsh), _p(dscale), _p(drot), _p(dopac), C.c_float(lambda_sh_sparsity))
    return dict(dL_dmeans2D=dmean2D, dL_dcolors=dcolor, dL_dopacity=dopac, dL_dmeans3D=dmean3D,
                dL_dcov3D=dcov3D, dL_dsh=dsh, dL_dscales=dscale, dL_drotations=drot, dL_dconic=dconic)


def preprocess_bwd_from(st, dL_dmeans2D, dL_dconic, dL_dcolors, lambda_sh_sparsity=0.0):
    """The per-Gaussian half of `backward` alone (backward.cu:177-434 as restated in orc_preprocess_bwd), fed GIVEN 2D-stage
    sums instead of the oracle's own: dL_dmeans2D [P,3], dL_dconic [P,4] (A, B, -, C), dL_dcolors [P,3].  What a test uses to
    separate the covariance chain's arithmetic from the rounding of its inputs: the chain amplifies a relative 1e-6 on the
    2D-stage sums of an anisotropic splat to 1e-4 .. 1e-3 on dL_drotations, so two correct fp32 pipelines whose blend stages
    sum in different orders disagree there, while the chain itself, fed the same numbers, must agree to rounding.
    (dL_dopacity is not an input of anything returned here.)"""
    L = lib()
    P, M, W, H = st["P"], st["M"], st["W"], st["H"]
    dmean2D = np.ascontiguousarray(dL_dmeans2D, dtype=np.float32).reshape(P, 3).copy()
    dconic = np.ascontiguousarray(dL_dconic, dtype=np.float32).reshape(P, 4).copy()
    dcolor = np.ascontiguousarray(dL_dcolors, dtype=np.float32).reshape(P, 3).copy()
    dopac = np.zeros((P, 1), np.float32)
    dmean3D, dcov3D = np.zeros((P, 3), np.float32), np.zeros((P, 6), np.float32)
    dsh, dscale, drot = np.zeros((P, M, 3), np.float32), np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32)
    if P:
        cov = st["cov3D_precomp"] if st["cov3D_precomp"] is not None else st["cov3D"]
        L.orc_preprocess_bwd(C.c_int(P), C.c_int(M), _p(st["degrees"]), _p(st["means3D"]), _p(st["radii"]),
                             _p(st["sh"]), _p(st["clamped"]), _p(st["scales"]), _p(st["rotations"]),
                             C.c_float(st["mod"]), _p(cov), _p(st["vm"]), _p(st["pm"]), _p(st["campos"]),
                             C.c_int(W), C.c_int(H), C.c_float(st["tan_fovx"]), C.c_float(st["tan_fovy"]),
                             _p(dmean2D), _p(st["conic_op"]), _p(dconic), _p(dmean3D), _p(dcolor), _p(dcov3D),
                             _p(dsh), _p(dscale), _p(drot), _p(dopac), C.c_float(lambda_sh_sparsity))
    return dict(dL_dmeans3D=dmean3D, dL_dcov3D=dcov3D, dL_dsh=dsh, dL_dscales=dscale, dL_drotations=drot)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def preprocess_bwd_f64(st, acc, lambda_sh_sparsity=0.0, fwd64=None, pure=False):
    """The per-Gaussian half of `backward_f64` alone: `acc` [P,9] float64 = the 2D-stage sums per Gaussian (dmean2D.x, .y
    in NDC units, dL/dconic A, HALF B, C, dL/d(activated opacity), dL/dcolour[3]) -> the chain's outputs in double."""
    L = lib()
    L.orc_f64_set_pure(C.c_int(1 if pure else 0))
    P, M, W, H = st["P"], st["M"], st["W"], st["H"]
    fw = fwd64 or {}
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    conic_op = _f64(fw.get("conic_op", st["conic_op"]))
    cov = _f64(fw.get("cov3D", st["cov3D_precomp"] if st["cov3D_precomp"] is not None else st["cov3D"]))
    dmean3D, dcov3D = np.zeros((P, 3), np.float64), np.zeros((P, 6), np.float64)
    dsh, dscale, drot = np.zeros((P, M, 3), np.float64), np.zeros((P, 3), np.float64), np.zeros((P, 4), np.float64)
    dopac = np.zeros((P, 1), np.float64)
    if P:
        L.orc_preprocess_bwd_f64(C.c_int(P), C.c_int(M), _p(st["degrees"]), _p(st["means3D"]), _p(st["radii"]),
                                 _p(st["sh"]), _p(st["clamped"]), _p(st["scales"]), _p(st["rotations"]),
                                 C.c_float(st["mod"]), _p(cov), _p(st["vm"]), _p(st["pm"]), _p(st["campos"]),
                                 C.c_int(W), C.c_int(H), C.c_float(st["tan_fovx"]), C.c_float(st["tan_fovy"]),
                                 _p(conic_op), _p(acc), _p(dmean3D), _p(dcov3D), _p(dsh), _p(dscale), _p(drot),
                                 _p(dopac), C.c_float(lambda_sh_sparsity))
    return dict(dL_dopacity=dopac, dL_dmeans3D=dmean3D, dL_dcov3D=dcov3D, dL_dsh=dsh, dL_dscales=dscale,
                dL_drotations=drot)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def backward_f64(st, dL_dout_color, lambda_sh_sparsity=0.0, fwd64=None, pure=False):
    """The same gradients as `backward`, evaluated in DOUBLE (oracle/backward_f64.c): the exact gradient of the function
    the forward evaluated, given the forward's fp32 state and its discrete decisions.  An independent statement (matrix
    calculus, transmittances by forward products), not a transcription of backward.cu -- the third party that says which
    of two fp32 evaluations of the covariance chain is the one that is off.  Returns float64 arrays under the same keys.
    fwd64 (pin test only): dict(xy, conic_op, colors, cov3D) of float64 arrays that replace the state's fp32 numbers.
    pure (pin test only): the pure derivative of the forward instead of the reference's conventions (fp32 focal lengths
    and clamp limits, 1 / (det^2 + 1e-7)); see backward_f64.c."""
    L = lib()
    P, W, H = st["P"], st["W"], st["H"]
    g = _f32(dL_dout_color)
    acc = np.zeros((P, 9), np.float64)
    if P:
        fw = fwd64 or {}
        feat = _f64(fw.get("colors", st["colors_precomp"] if st["colors_precomp"] is not None else st["rgb"]))
        xy, conic_op = _f64(fw.get("xy", st["xy"])), _f64(fw.get("conic_op", st["conic_op"]))
        L.orc_blend_bwd_f64(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["bg"]),
                            _p(xy), _p(conic_op), _p(feat), _p(st["n_contrib"]), _p(g), _p(acc))
    out = preprocess_bwd_f64(st, acc, lambda_sh_sparsity, fwd64, pure)
    dmean2D = np.zeros((P, 3), np.float64)
    dmean2D[:, :2] = acc[:, 0:2]
    dconic = np.zeros((P, 4), np.float64)
    dconic[:, [0, 1, 3]] = acc[:, 2:5]
    out.update(dL_dmeans2D=dmean2D, dL_dcolors=acc[:, 6:9].copy(), dL_dconic=dconic)
    return out


def sh_backward(means3D, sh, degrees, campos, clamped, dL_dcolor):
    """The SH part of the per-Gaussian backward alone (backward.cu:20-172 computeColorFromSH as restated in
    orc_preprocess_bwd): dL/dsh [P,M,3] and the view-direction part of dL/dmeans [P,3] for a given dL/dcolor, with every
    other upstream gradient zero and a dummy, well-conditioned camera (points ~10 units in front)."""
    L = lib()
    means3D, sh, campos, dcolor = _f32(means3D), _f32(sh), _f32(campos), _f32(dL_dcolor)
    degrees = _i32(degrees)
    P, M = sh.shape[0], sh.shape[1]
    clamped = np.ascontiguousarray(clamped, dtype=np.uint8)
    vm = np.eye(4, dtype=np.float32)
    vm[3, 2] = 10.0
    pm = vm.copy()
    radii = np.ones(P, np.int32)
    scales = np.full((P, 3), 0.01, np.float32)
    rots = np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1))
    cov = np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32), (P, 1))
    conic_op = np.tile(np.array([1, 0, 1, 0.5], np.float32), (P, 1))
    z3, z4, z1 = np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32), np.zeros((P, 1), np.float32)
    dmean3D, dcov3D = np.zeros((P, 3), np.float32), np.zeros((P, 6), np.float32)
    dsh, dscale, drot = np.zeros((P, M, 3), np.float32), np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32)
    L.orc_preprocess_bwd(C.c_int(P), C.c_int(M), _p(degrees), _p(means3D), _p(radii), _p(sh), _p(clamped), _p(scales),
                         _p(rots), C.c_float(1.0), _p(cov), _p(vm), _p(pm), _p(campos), C.c_int(64), C.c_int(64),
                         C.c_float(1.0), C.c_float(1.0), _p(z3), _p(conic_op), _p(z4), _p(dmean3D), _p(dcolor),
                         _p(dcov3D), _p(dsh), _p(dscale), _p(drot), _p(z1), C.c_float(0.0))
    return dsh, dmean3D


def mark_visible(means3D, viewmatrix):
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    L.orc_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix)), _p(out))
    return out.astype(bool)


def higher_msb(n):
    return int(lib().orc_higher_msb(C.c_uint32(n)))
