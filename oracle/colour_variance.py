"""Oracle of the next-tier operator `_C.calculate_colours_variance`
(DGR/reduced_3dgs.cu:41-203 + DGR/reduced_3dgs/sh_culling.cu:6-90): numpy fp32 restatement of the reference's
host loop over cameras, built on the oracle's counter-mode forward.  TEST INFRASTRUCTURE ONLY.

Reference quirks reproduced:
  * `auto mean_old = mean;` aliases the tensor (reduced_3dgs.cu:185), so the variance update uses the UPDATED
    mean in both factors: variance += w * (colour - mean_new)^2;
  * truncated-colour slots above a Gaussian's own degree stay 0, including the "full colour" slot
    [max_sh_deg] for Gaussians of lower degree (sh_culling.cu:21-54);
  * Gaussians never seen end with 0/0 = NaN in the returned distances / variance.
The reference's kernel hard-codes 4 slots per Gaussian (only valid for max_sh_deg == 3); this restatement uses
max_sh_deg + 1 slots, identical for max_sh_deg == 3."""
import ctypes as C

import numpy as np

from . import oracle as orc


def calculate_colours_variance(cam_positions, means3D, opacity, scales, rotations, cam_viewmatrices,
                               cam_projmatrices, tan_fovxs, tan_fovys, image_height, image_width, sh, degrees,
                               max_sh_deg):
    L = orc.lib()
    f32 = np.float32
    means3D = np.ascontiguousarray(means3D, f32)
    sh = np.ascontiguousarray(sh, f32)
    degrees_i = np.ascontiguousarray(degrees, np.int32).reshape(-1)
    P, M = means3D.shape[0], sh.shape[1]
    S = max_sh_deg + 1
    accum = np.zeros((P, max_sh_deg), f32)
    wSum = np.zeros((P, 1), f32)
    mean = np.zeros((P, 1, 3), f32)
    variance = np.zeros((P, 1, 3), f32)
    for i in range(len(cam_positions)):
        H, W = int(image_height[i]), int(image_width[i])
        out = orc.forward(np.zeros(3, f32), means3D, None, opacity, scales, rotations, 1.0, None,
                          cam_viewmatrices[i], cam_projmatrices[i], float(tan_fovxs[i]), float(tan_fovys[i]), H, W, sh,
                          degrees, cam_positions[i], counter_mode=True)
        present = out["radii"] > 0
        touched = out["touched_pixels"].astype(f32).reshape(P, 1)
        w = out["transmittance"].reshape(P, 1) / np.maximum(touched, f32(1.0))
        wSum = wSum + w
        colours = np.zeros((P, S, 3), f32)
        campos = np.ascontiguousarray(cam_positions[i], f32)
        L.orc_truncated_colours(C.c_int(P), C.c_int(M), C.c_int(S), orc._p(degrees_i), orc._p(means3D), orc._p(campos),
                                orc._p(sh), orc._p(colours))
        colours[~present] = 0
        full = colours[:, max_sh_deg:max_sh_deg + 1]
        for cur in range(max_sh_deg):
            d = np.sqrt(((full - colours[:, cur:cur + 1]) ** 2).sum(2, dtype=f32)).astype(f32)
            d[np.isnan(d)] = 0
            accum[:, cur:cur + 1] = accum[:, cur:cur + 1] + w * d
        with np.errstate(invalid="ignore", divide="ignore"):
            coeff = (w / wSum).astype(f32)
        coeff[np.isnan(coeff)] = 0
        mean_new = mean.copy()
        mean_new[present] = mean[present] + coeff[present].reshape(-1, 1, 1) * (full[present] - mean[present])
        variance[present] = variance[present] + w[present].reshape(-1, 1, 1) * (full[present] - mean_new[present]) * (
            full[present] - mean_new[present])
        mean = mean_new
    with np.errstate(invalid="ignore", divide="ignore"):
        return (accum / wSum).astype(f32), (variance / wSum.reshape(-1, 1, 1)).astype(f32), mean
