"""The reference's own Python call sites of the operator modules, driven on the CPU against SIGNATURE-CHECKING stubs
("a CPU-container variant that at least drives GaussianModel.save_ply -> load_ply -> produce_clusters call
signatures against stubs so an API drift in _C is caught where the reference exists").

tests/test_reference_loop.py runs the reference's whole loop on the real operators, but needs a GPU AND a reference checkout
and so runs nowhere in this project; this file needs only the checkout (the authoring container) and skips on the GPU box.
Every operator the reference imports from `diff_gaussian_rasterization._C` / `simple_knn._C` is replaced by a stub that
  (1) binds the call's arguments to the signature of THIS repository's operator (inspect.signature(...).bind): a positional
      argument added, dropped or reordered on either side fails here;
  (2) checks the argument shapes / dtypes the operator's C-ABI wrapper requires;
  (3) returns tensors of the shapes and dtypes the real operator returns, so that the reference's code after the call runs.
What runs, unmodified reference code: GaussianModel.produce_clusters (kmeans_cuda x 20) -> save_ply(quantised) -> load_ply ->
apply_clustering; GaussianModel.cull_sh_bands (calculate_colours_variance x 2); Scene.calculate_redundancy_metric
(find_minimum_projected_pixel_size, distIndex2, sphere_ellipsoid_intersection, allocate_minimum_redundancy_value);
gaussian_renderer.render (GaussianRasterizer -> _C.rasterize_gaussians).  `torch.Tensor.cuda` and the `device="cuda"` of the
tensor factories are made no-ops (that code hard-codes them); nothing of the reference is stored in this repository."""
import inspect
import math
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PKG = os.path.join(ROOT, "reduced-3dgs_amd")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")),
                               reason="reference tree not present (authoring container only)")
sys.path.insert(0, PKG)

FACTORIES = ("zeros", "ones", "empty", "full", "tensor", "randint", "rand", "randn", "arange", "zeros_like", "ones_like",
             "empty_like", "full_like", "eye", "linspace")


@pytest.fixture()
def ref_cpu(monkeypatch):
    """The reference's modules importable, with this repository's packages in front of its submodules, on a machine without
    a GPU: `.cuda()` and `device="cuda"` become no-ops."""
    import torch
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(PKG)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    for name in FACTORIES:
        real = getattr(torch, name)

        def make(real):
            def f(*a, **k):
                if str(k.get("device", "")).startswith("cuda"):
                    k = dict(k, device="cpu")
                return real(*a, **k)
            return f
        monkeypatch.setattr(torch, name, make(real))
    if not hasattr(np, "cast"):   # np.cast[np.float16](..) (gaussian_model.py:269) left NumPy in 2.0
        class _Cast:
            def __getitem__(self, dtype):
                return lambda a: np.asarray(a, dtype=dtype)
        monkeypatch.setattr(np, "cast", _Cast(), raising=False)
    mods = ("scene", "utils", "gaussian_renderer", "arguments")
    for m in [k for k in sys.modules if k.split(".")[0] in mods]:
        monkeypatch.delitem(sys.modules, m)
    yield
    for m in [k for k in sys.modules if k.split(".")[0] in mods]:
        sys.modules.pop(m, None)


def real_signature(module, name):
    """Signature of this repository's operator (its private keyword-only extras removed)."""
    sig = inspect.signature(getattr(module, name))
    return sig.replace(parameters=[p for p in sig.parameters.values() if not p.name.startswith("_")])


calls = {}


def stub(module, name, check, result):
    sig = real_signature(module, name)

    def f(*a, **k):
        bound = sig.bind(*a, **k)          # (1): the call fits THIS repository's operator
        bound.apply_defaults()
        check(**bound.arguments)           # (2): shapes / dtypes the C-ABI wrapper requires
        calls[name] = calls.get(name, 0) + 1
        return result(**bound.arguments)   # (3): what the real operator returns
    f.__name__ = name
    return f


def _model(gm, P=600, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    m = gm.GaussianModel(3)
    m.active_sh_degree = 3
    m._degrees = torch.randint(0, 4, (P, 1), generator=g, dtype=torch.int32)
    m._xyz = torch.randn(P, 3, generator=g) * 0.5 + torch.tensor([0.0, 0.0, 4.0])
    m._features_dc = torch.randn(P, 1, 3, generator=g)
    keep = (torch.arange(15)[None, :] < ((m._degrees + 1) ** 2 - 1)).unsqueeze(-1)
    m._features_rest = torch.randn(P, 15, 3, generator=g) * keep
    m._opacity = torch.randn(P, 1, generator=g)
    m._scaling = torch.randn(P, 3, generator=g) - 3.0
    m._rotation = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    return m


@needs_ref
def test_reference_call_sites_fit_this_repositorys_operators(ref_cpu, tmp_path):
    import torch
    from diff_gaussian_rasterization import _C as C_        # the ctypes module loads without a GPU; its operators refuse CPU tensors
    from simple_knn import _C as K_
    calls.clear()
    f32, i32 = torch.float32, torch.int32

    def is_t(t, dtype, *shape):
        assert isinstance(t, torch.Tensor) and t.dtype == dtype, (t.dtype if isinstance(t, torch.Tensor) else type(t), dtype)
        assert len(shape) == 0 or tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)

    # ---- stubs of the operators, installed where the reference's modules will import them from --------------------------
    def km_check(values, centers, tol, max_iterations):
        is_t(values, f32, values.shape[0], 1)
        assert centers.dtype == f32 and centers.dim() == 1 and centers.numel() <= 256
        assert isinstance(tol, float) and int(max_iterations) == 500

    def km_result(values, centers, tol, max_iterations):
        ids = (values.view(-1, 1) - centers.view(1, -1)).abs().argmin(dim=1).to(i32).view(-1, 1)
        return ids, centers.clone()

    def cv_check(cam_positions, means3D, opacity, scales, rotations, cam_viewmatrices, cam_projmatrices, tan_fovxs, tan_fovys,
                 image_height, image_width, sh, degrees, max_sh_deg):
        P, n = means3D.shape[0], cam_positions.shape[0]
        is_t(cam_positions, f32, n, 3), is_t(means3D, f32, P, 3), is_t(opacity, f32, P, 1), is_t(scales, f32, P, 3)
        is_t(rotations, f32, P, 4), is_t(cam_viewmatrices, f32, n, 4, 4), is_t(cam_projmatrices, f32, n, 4, 4)
        is_t(tan_fovxs, f32, n), is_t(tan_fovys, f32, n), is_t(image_height, i32, n), is_t(image_width, i32, n)
        is_t(sh, f32, P, 16, 3), is_t(degrees, i32, P, 1)
        assert int(max_sh_deg) == 3

    def cv_result(means3D, max_sh_deg, **_):
        P = means3D.shape[0]
        g = torch.Generator().manual_seed(1)
        return (torch.rand(P, int(max_sh_deg), generator=g) * 0.02, torch.rand(P, 1, 3, generator=g) * 1e-4,
                torch.rand(P, 1, 3, generator=g))

    def px_check(w2ndc_transforms, w2ndc_transforms_inverse, means3D, image_height, image_width):
        n = w2ndc_transforms.shape[0]
        is_t(w2ndc_transforms, f32, n, 4, 4), is_t(w2ndc_transforms_inverse, f32, n, 4, 4)
        is_t(means3D, f32, means3D.shape[0], 3), is_t(image_height, i32, n), is_t(image_width, i32, n)

    def se_check(means3D, scales, rotations, neighbours_indices, sphere_radius, knn):
        P = means3D.shape[0]
        is_t(means3D, f32, P, 3), is_t(scales, f32, P, 3), is_t(rotations, f32, P, 4)
        is_t(neighbours_indices, i32, P, int(knn)), is_t(sphere_radius, f32, P, 1)

    def mr_check(redundancy_values, neighbours_indices, intersection_mask, knn):
        P = redundancy_values.shape[0]
        is_t(redundancy_values, i32, P, 1), is_t(neighbours_indices, i32, P, int(knn))
        is_t(intersection_mask, torch.bool, P, int(knn))

    def rg_check(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                 tan_fovx, tan_fovy, image_height, image_width, sh, degrees, campos, prefiltered, debug):
        P = means3D.shape[0]
        is_t(background, f32, 3), is_t(means3D, f32, P, 3), is_t(opacity, f32, P, 1), is_t(scales, f32, P, 3)
        is_t(rotations, f32, P, 4), is_t(viewmatrix, f32, 4, 4), is_t(projmatrix, f32, 4, 4), is_t(sh, f32, P, 16, 3)
        is_t(degrees, i32, P, 1), is_t(campos, f32, 3)
        assert colors.numel() == 0 and cov3D_precomp.numel() == 0          # "not provided" is an empty tensor
        assert isinstance(image_height, int) and isinstance(image_width, int) and isinstance(prefiltered, bool)

    def rg_result(means3D, image_height, image_width, **_):
        P = means3D.shape[0]
        z = torch.zeros
        return (7, z(3, image_height, image_width), torch.ones(P, dtype=i32), z(16, dtype=torch.uint8), z(16, dtype=torch.uint8),
                z(16, dtype=torch.uint8))

    def knn_check(points, K):
        is_t(points, f32, points.shape[0], 3)
        assert int(K) == 30

    def knn_result(points, K):
        P = points.shape[0]
        d = torch.cdist(points, points)
        d.fill_diagonal_(float("inf"))
        dist, idx = d.topk(int(K), largest=False)
        return dist.reshape(-1), idx.to(i32).reshape(-1)

    stubs = {
        "kmeans_cuda": stub(C_, "kmeans_cuda", km_check, km_result),
        "calculate_colours_variance": stub(C_, "calculate_colours_variance", cv_check, cv_result),
        "find_minimum_projected_pixel_size": stub(C_, "find_minimum_projected_pixel_size", px_check,
                                                  lambda means3D, **_: torch.full((means3D.shape[0], 1), 0.01)),
        "sphere_ellipsoid_intersection": stub(C_, "sphere_ellipsoid_intersection", se_check,
                                              lambda means3D, knn, **_: (torch.ones(means3D.shape[0], 1, dtype=i32),
                                                                         torch.zeros(means3D.shape[0], int(knn), dtype=torch.bool))),
        "allocate_minimum_redundancy_value": stub(C_, "allocate_minimum_redundancy_value", mr_check,
                                                  lambda redundancy_values, **_: (redundancy_values.clone(),)),
        "rasterize_gaussians": stub(C_, "rasterize_gaussians", rg_check, rg_result),
    }
    fake_c = types.ModuleType("diff_gaussian_rasterization._C")
    for n in dir(C_):
        if not n.startswith("__"):
            setattr(fake_c, n, stubs.get(n, getattr(C_, n)))
    fake_k = types.ModuleType("simple_knn._C")
    fake_k.distCUDA2 = K_.distCUDA2
    fake_k.distIndexQ = K_.distIndexQ
    fake_k.distIndex2 = stub(K_, "distIndex2", knn_check, knn_result)
    import diff_gaussian_rasterization as dgr
    import simple_knn
    old = (sys.modules["diff_gaussian_rasterization._C"], sys.modules["simple_knn._C"], dgr._C, simple_knn._C)
    sys.modules["diff_gaussian_rasterization._C"], sys.modules["simple_knn._C"] = fake_c, fake_k
    dgr._C, simple_knn._C = fake_c, fake_k
    try:
        import gaussian_renderer as gr
        import scene as sc
        import scene.gaussian_model as gm
        from scene.cameras import Camera
        assert gm.kmeans_cuda is stubs["kmeans_cuda"] and sc.find_minimum_projected_pixel_size is stubs["find_minimum_projected_pixel_size"]

        # ---- compress.py's path: produce_clusters -> save_ply(quantised) -> load_ply -> apply_clustering ----------------
        m = _model(gm)
        P = m._xyz.shape[0]
        m.produce_clusters(store_dict_path=None)
        assert calls["kmeans_cuda"] == 20 and len(m._codebook_dict) == 20
        assert m._codebook_dict["scaling"].ids.dtype == torch.uint8 and tuple(m._codebook_dict["scaling"].ids.shape) == (P, 3)
        path = str(tmp_path / "point_cloud_quantised.ply")
        m.save_ply(path, quantised=True, half_float=True)
        back = gm.GaussianModel(3)
        back.load_ply(path, half_float=True, quantised=True)
        assert back._xyz.shape[0] == P and back._features_rest.shape == (P, 15, 3)
        m.apply_clustering(m._codebook_dict)
        assert m._opacity.shape == (P, 1) and m._rotation.shape == (P, 4)

        # ---- train.py's culling / pruning paths --------------------------------------------------------------------------
        W, H = 64, 48
        fov = 2 * math.atan(W / (2 * 60.0))
        cams = [Camera(colmap_id=i, R=np.eye(3), T=np.array([0.1 * i, 0.0, 0.0]), FoVx=fov, FoVy=fov * H / W,
                       image=torch.zeros(3, H, W), gt_alpha_mask=None, image_name=f"c{i}", uid=i, data_device="cpu")
                for i in range(3)]
        m = _model(gm, seed=2)
        m.cull_sh_bands(cams, threshold=0.04, std_threshold=0.01)
        assert calls["calculate_colours_variance"] == 2
        holder = types.SimpleNamespace(gaussians=m, getTrainCameras=lambda: cams)
        red, cube = sc.Scene.calculate_redundancy_metric(holder, pixel_scale=1.0, num_neighbours=30)
        assert red.shape == (P, 1) and cube.shape == (P, 1)
        for n in ("find_minimum_projected_pixel_size", "distIndex2", "sphere_ellipsoid_intersection",
                  "allocate_minimum_redundancy_value"):
            assert calls.get(n) == 1, n

        # ---- gaussian_renderer.render -> GaussianRasterizer -> _C.rasterize_gaussians ------------------------------------
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        out = gr.render(cams[0], m, pipe, torch.zeros(3), lambda_sh_sparsity=0.0)
        assert calls["rasterize_gaussians"] == 1 and out["render"].shape == (3, H, W) and out["radii"].shape == (P,)
    finally:
        sys.modules["diff_gaussian_rasterization._C"], sys.modules["simple_knn._C"], dgr._C, simple_knn._C = old
