"""diff_gaussian_rasterization -- MI355X-native drop-in for the rasterizer package of
graphdeco-inria/reduced-3dgs (submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py).

Public surface kept identical so that gaussian_renderer.render(), train.py and render.py run unmodified:
`GaussianRasterizationSettings` (12 fields, same order, :169-181), `GaussianRasterizer(raster_settings)`
with `.forward(...)` / `.markVisible(...)` (:183-234), `rasterize_gaussians(...)` and the autograd
function `_RasterizeGaussians` (:21-167), plus the `_C` operator module.

Put `reduced-3dgs_amd/` on PYTHONPATH (before or instead of the CUDA submodule) to switch a reference
checkout over.  Behavioural notes:
  * the reference hard-codes debug=True in the forward call (:85), i.e. a device synchronisation after
    every stage; here `raster_settings.debug` is honoured.  debug=False (what render() passes) takes the
    asynchronous path: nothing waits for num_rendered, a pass is one hipGraph launch, and `ctx.num_rendered`
    is an int-like `_C.NumRendered` that is only fetched from the device if somebody looks at it;
    debug=True takes the exact-size path with a synchronisation after every stage;
  * gradients are bit-reproducible: the backward has no float atomics (the reference issues one per
    (pixel, Gaussian, component), so its low bits depend on scheduling).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    """Host snapshot of an argument tuple, written out if the native call fails in debug mode."""
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _call_native(fn, args, debug, dump_name):
    if not debug:
        return fn(*args)
    snapshot = cpu_deep_copy_tuple(args)  # taken before the call so a crash cannot corrupt it
    try:
        return fn(*args)
    except Exception:
        torch.save(snapshot, dump_name)
        print(f"\nAn error occured in the rasterizer. Inputs were written to {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd boundary of the tile rasterizer.  Inputs (positional, as the reference):
    means3D, means2D, sh, degrees, colors_precomp, opacities (raw), scales (activated), rotations (unit),
    cov3Ds_precomp, raster_settings, lambda_sh_sparsity  ->  (color[3,H,W], radii[P])."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, degrees, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, lambda_sh_sparsity):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, degrees,
                rs.campos, rs.prefiltered, rs.debug)
        _C.hint_next_forward(any(ctx.needs_input_grad))   # rendering under no_grad: nothing is kept for a backward's sake
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _call_native(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.lambda_sh_sparsity = lambda_sh_sparsity
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, degrees)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zeros_like(radii) fill per backward for the non-differentiable output
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         degrees) = ctx.saved_tensors
        if grad_out_color is None:  # the image did not take part in the loss
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=means3D.dtype,
                                         device=means3D.device)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, degrees, rs.campos,
                geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, ctx.lambda_sh_sparsity, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_native(_C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump")
        # one slot per forward input; None for degrees, raster_settings, lambda_sh_sparsity
        return (grad_means3D, grad_means2D, grad_sh, None, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None, None)


def rasterize_gaussians(means3D, means2D, sh, degrees, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, lambda_sh_sparsity):
    return _RasterizeGaussians.apply(means3D, means2D, sh, degrees, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, lambda_sh_sparsity)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: which positions pass the near-plane test of this camera (view-space z > 0.2)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, degrees=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, lambda_sh_sparsity=0.):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # absent optional inputs travel as empty tensors (-> NULL at the C ABI), like the reference
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, degrees, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings, lambda_sh_sparsity)
