"""Dense PyTorch (autograd) restatement of the rasterizer for TINY problems.

TEST INFRASTRUCTURE ONLY.  Purpose: an independent, differentiable statement of the
forward maths whose fp64 autograd gradients pin the C oracle's
hand-written backward (oracle/raster_oracle.c, which follows DGR/cuda_rasterizer/backward.cu).
The default path evaluates every (pixel, Gaussian) pair densely -- O(N*P) memory -- so keep P <= ~2000 and
images <= ~64x64 there; `tiled=True` blends tile by tile against each tile's own Gaussians (same arithmetic), which
is what makes it usable as the "PyTorch CPU autograd reference render" BASELINE.json configs[0] names
(10k Gaussians, 400x400: about a second per pass) -- bench.py times that as `cpu_baseline`.

Reference quirks reproduced on purpose (so that autograd == the reference's analytic backward):
  * alpha = min(0.99, o*G) is straight-through in the backward (backward.cu:537,576);
  * colour clamp at 0 kills the gradient (backward.cu:29-34) -- relu does the same;
  * radius / tile rect / depth order / termination are integer decisions, not differentiated;
  * quaternions are used as given (no normalisation inside, forward.cu:216).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
TILE = 16


def sh_basis(x, y, z):
    """[P,16] real SH basis, polynomial form of DGR/cuda_rasterizer/forward.cu:115-148."""
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    one = torch.ones_like(x)
    return torch.stack([
        SH_C0 * one, -SH_C1 * y, SH_C1 * z, -SH_C1 * x,
        SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy),
        SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
        SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
        SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)], dim=1)


def quat_to_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def render(means3D, opacity_raw, scales, rotations, sh, degrees, viewmatrix, projmatrix, campos, bg,
           W, H, tan_fovx, tan_fovy, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None,
           lambda_sh_sparsity=0.0, tiled=False, geo_out=None):
    """Returns (color[3,H,W], radii[P], sh_sparsity_loss).  All float tensors share one dtype
    (use float64 for gradient ground truth).  `viewmatrix`/`projmatrix` are the transposed
    (row-vector) matrices exactly as the reference passes them."""
    dt = means3D.dtype
    P = means3D.shape[0]
    V, F = viewmatrix.to(dt), projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    t = torch.cat([means3D, ones], 1) @ V  # row-vector convention == transformPoint4x3
    h = torch.cat([means3D, ones], 1) @ F
    vis = t[:, 2] > 0.2
    pw = 1.0 / (h[:, 3] + 1e-7)
    mx = ((h[:, 0] * pw + 1.0) * W - 1.0) * 0.5
    my = ((h[:, 1] * pw + 1.0) * H - 1.0) * 0.5
    if cov3D_precomp is None:
        R = quat_to_R(rotations)
        S = scales * scale_modifier
        L = R * S[:, None, :]
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            1).reshape(-1, 3, 3)
    fx, fy = W / (2.0 * tan_fovx), H / (2.0 * tan_fovy)
    tz = t[:, 2]
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz  # clamp -> zero gradient outside, like x_grad_mul
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    # the reference keeps d(t.x_clamped)/d(t.z) = clamp value only through J's explicit tz terms: its
    # backward differentiates J entries w.r.t. (t.x, t.y, t.z) treating the clamped t.x, t.y as the
    # variables (backward.cu:295-297), with x_grad_mul zeroing the t.x / t.y paths when clamped.
    tx_var = torch.where((t[:, 0] / tz).abs() > limx, txc.detach(), t[:, 0])
    ty_var = torch.where((t[:, 1] / tz).abs() > limy, tyc.detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx_var) / (tz * tz),
                     zero, fy / tz, -(fy * ty_var) / (tz * tz)], 1).reshape(-1, 2, 3)
    Rw = V[:3, :3].t()
    A = J @ Rw
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c_ = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    vis = vis & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    cA, cB, cC = c_ / det_safe, -b / det_safe, a / det_safe
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def tr(v):
        return torch.trunc(v).to(torch.int64)

    mxd, myd = mx.detach(), my.detach()
    rminx = tr((mxd - radius) / TILE).clamp(0, gx)
    rminy = tr((myd - radius) / TILE).clamp(0, gy)
    rmaxx = tr((mxd + radius + TILE - 1) / TILE).clamp(0, gx)
    rmaxy = tr((myd + radius + TILE - 1) / TILE).clamp(0, gy)
    vis = vis & (((rmaxx - rminx) * (rmaxy - rminy)) != 0)
    radii = torch.where(vis, radius.to(torch.int32), torch.zeros(P, dtype=torch.int32))
    # colour
    if colors_precomp is None:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        Y = sh_basis(d[:, 0], d[:, 1], d[:, 2])
        K = (degrees.reshape(-1).to(torch.int64) + 1) ** 2
        kmask = (torch.arange(16)[None, :] < K[:, None]).to(dt)
        Msh = sh.shape[1]
        rgb = torch.clamp(((Y * kmask)[:, :Msh, None] * sh).sum(1) + 0.5, min=0.0)
        sparsity = torch.zeros((), dtype=dt)
        if lambda_sh_sparsity != 0.0:
            nvis = int(vis.sum())
            mult = lambda_sh_sparsity / (nvis * 15 * 3)
            km = kmask[:, :Msh].clone()
            km[:, 0] = 0
            sparsity = mult * (sh.abs() * km[:, :, None] * vis[:, None, None].to(dt)).sum()
    else:
        rgb = colors_precomp
        sparsity = torch.zeros((), dtype=dt)
    o = torch.sigmoid(opacity_raw.reshape(-1))
    # depth order: stable sort on depth, ties by index (rasterizer_impl.cu:106-117,468)
    depth32 = t[:, 2].detach().to(torch.float32)
    order = torch.sort(depth32, stable=True).indices
    order = order[vis[order]]
    geo = dict(mx=mx, my=my, cA=cA, cB=cB, cC=cC, o=o, rgb=rgb, rminx=rminx, rminy=rminy, rmaxx=rmaxx, rmaxy=rmaxy)
    if geo_out is not None:   # this forward's per-Gaussian quantities (for oracle.backward_f64's pin test)
        geo_out.update(xy=torch.stack([mx, my], 1).detach().numpy(),
                       conic_op=torch.stack([cA, cB, cC, o], 1).detach().numpy(), colors=rgb.detach().numpy(),
                       cov3D=torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2],
                                          Sigma[:, 2, 2]], 1).detach().numpy())
    if tiled:
        # Tile by tile, each tile against the Gaussians whose rect covers it (what the binning hands the blend kernel):
        # O(256 * n_tile) work per tile instead of O(N * P), which makes BASELINE.json configs[0] (10k Gaussians,
        # 400x400) tractable on a CPU.  Same arithmetic as the dense path.
        out = torch.empty(H, W, 3, dtype=dt)
        ro_x0, ro_x1, ro_y0, ro_y1 = rminx[order], rmaxx[order], rminy[order], rmaxy[order]
        for ty in range(gy):
            rows = (ro_y0 <= ty) & (ro_y1 > ty)
            for tx in range(gx):
                g = order[rows & (ro_x0 <= tx) & (ro_x1 > tx)]
                y0, y1, x0, x1 = ty * TILE, min(H, (ty + 1) * TILE), tx * TILE, min(W, (tx + 1) * TILE)
                ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
                out[y0:y1, x0:x1] = _blend(xs.reshape(-1), ys.reshape(-1), g, geo, bg, dt).reshape(y1 - y0, x1 - x0, 3)
        return out.permute(2, 0, 1), radii, sparsity
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    out = _blend(xs.reshape(-1), ys.reshape(-1), order, geo, bg, dt)
    return out.t().reshape(3, H, W), radii, sparsity


def _blend(xs, ys, g, geo, bg, dt):
    """Front-to-back alpha blend of the pixels (xs, ys) over the depth-ordered Gaussians g -> [n_pix, 3]."""
    px, py = xs.to(dt), ys.to(dt)
    tx_pix, ty_pix = xs // TILE, ys // TILE
    in_rect = ((tx_pix[:, None] >= geo["rminx"][g][None, :]) & (tx_pix[:, None] < geo["rmaxx"][g][None, :]) &
               (ty_pix[:, None] >= geo["rminy"][g][None, :]) & (ty_pix[:, None] < geo["rmaxy"][g][None, :]))
    dx = geo["mx"][g][None, :] - px[:, None]
    dy = geo["my"][g][None, :] - py[:, None]
    power = -0.5 * (geo["cA"][g][None, :] * dx * dx + geo["cC"][g][None, :] * dy * dy) - geo["cB"][g][None, :] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    araw = geo["o"][g][None, :] * G
    alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()  # straight-through min(0.99, .)
    valid = in_rect & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    Tincl = torch.cumprod(one_m, dim=1)
    stop = (torch.cumsum((valid & (Tincl.detach() < 1e-4)).to(torch.int64), dim=1) > 0)
    a_fin = torch.where(stop, torch.zeros_like(a_eff), a_eff)
    Tincl2 = torch.cumprod(1.0 - a_fin, dim=1)
    Texcl2 = torch.cat([torch.ones(Tincl2.shape[0], 1, dtype=dt), Tincl2[:, :-1]], 1)
    w = a_fin * Texcl2
    C = w @ geo["rgb"][g]
    Tend = Tincl2[:, -1] if Tincl2.shape[1] else torch.ones(px.shape[0], dtype=dt)
    return C + Tend[:, None] * bg.to(dt)[None, :]
