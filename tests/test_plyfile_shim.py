"""`reduced-3dgs_amd/plyfile.py` -- the minimal plyfile-compatible PLY codec.

1. format-level known answers written out from the PLY description (header text, row bytes);
2. round trips over the eight number types, three encodings, empty elements and list properties;
3. the reference's own save/load code (`scene/gaussian_model.py:239-311, 398-483`, `scene/dataset_readers.py:107-130`)
   run UNMODIFIED on top of this module, `simple_knn` and `diff_gaussian_rasterization` of this repository -- in the
   authoring container only (the reference tree is not present on the GPU box); `torch.Tensor.cuda` is made a no-op
   because that code hard-codes `.cuda()`.  Parity of the codec itself is unpinned against the real `plyfile`
   package (not installed here); what is pinned is the reference's use of the API."""
import io
import os
import struct
import sys
from collections import OrderedDict

import numpy as np
import pytest

import plyfile
from plyfile import PlyData, PlyElement

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def test_this_is_the_repo_module():
    assert os.path.dirname(os.path.abspath(plyfile.__file__)) == os.path.join(os.path.dirname(HERE), "reduced-3dgs_amd")


def test_known_answer_bytes():
    a = np.zeros(2, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("n", "i2")])
    a["x"], a["y"], a["z"] = [1.0, -2.5], [0.0, 4.0], [3.0, 0.5]
    a["red"], a["n"] = [7, 255], [-3, 12]
    buf = io.BytesIO()
    PlyData([PlyElement.describe(a, "vertex_0")]).write(buf)
    header = (b"ply\nformat binary_little_endian 1.0\nelement vertex_0 2\nproperty float x\nproperty float y\n"
              b"property float z\nproperty uchar red\nproperty short n\nend_header\n")
    rows = struct.pack("<fffBh", 1.0, 0.0, 3.0, 7, -3) + struct.pack("<fffBh", -2.5, 4.0, 0.5, 255, 12)
    assert buf.getvalue() == header + rows
    big = io.BytesIO()
    PlyData([PlyElement.describe(a, "vertex_0")], byte_order=">").write(big)
    assert big.getvalue().endswith(struct.pack(">fffBh", 1.0, 0.0, 3.0, 7, -3) + struct.pack(">fffBh", -2.5, 4.0, 0.5, 255, 12))
    assert b"format binary_big_endian 1.0" in big.getvalue()


@pytest.mark.parametrize("text,order", [(False, "<"), (False, ">"), (True, "=")])
def test_round_trip_all_types(text, order, tmp_path):
    rng = np.random.default_rng(0)
    n = 257
    dt = [("c", "i1"), ("uc", "u1"), ("s", "i2"), ("us", "u2"), ("i", "i4"), ("ui", "u4"), ("f", "f4"), ("d", "f8")]
    a = np.zeros(n, dtype=dt)
    for name, kind in dt:
        if kind[0] == "f":
            a[name] = rng.normal(0, 1e3, n).astype(kind)
        else:
            info = np.iinfo(kind)
            a[name] = rng.integers(info.min, info.max, n, endpoint=True).astype(kind)
    empty = np.zeros(0, dtype=[("x", "f4")])
    codebook = np.zeros(256, dtype=[("opacity", "i2"), ("scaling", "i2")])
    codebook["opacity"] = rng.normal(0, 1, 256).astype(np.float16).view(np.int16)   # half bit-cast, as the reference does
    path = tmp_path / "t.ply"
    PlyData([PlyElement.describe(a, "vertex"), PlyElement.describe(empty, "nothing"),
             PlyElement.describe(codebook, "codebook_centers")], text=text, byte_order=order,
            comments=["made by the test"]).write(str(path))
    back = PlyData.read(str(path))
    assert [e.name for e in back.elements] == ["vertex", "nothing", "codebook_centers"]
    assert back.comments == ["made by the test"] and back.text == text
    assert "vertex" in back and "face" not in back and "uc" in back["vertex"] and "q" not in back["vertex"]
    assert back["vertex"].count == n and back.elements[1].count == 0 and len(back) == 3
    for name, _ in dt:
        assert np.array_equal(back["vertex"][name], a[name]), name
        assert back["vertex"][name].dtype == a[name].dtype
    assert np.array_equal(back.elements[-1]["opacity"].view(np.float16), codebook["opacity"].view(np.float16))
    with pytest.raises(KeyError):
        back["face"]


def test_list_properties_and_errors(tmp_path):
    faces = np.empty(2, dtype=[("vertex_indices", "O"), ("flag", "u1")])
    faces["vertex_indices"][0] = np.array([0, 1, 2], "i4")
    faces["vertex_indices"][1] = np.array([2, 3, 4, 5], "i4")
    faces["flag"] = [1, 0]
    for text in (False, True):
        p = tmp_path / f"f{int(text)}.ply"
        PlyData([PlyElement.describe(faces, "face")], text=text).write(str(p))
        assert b"property list uchar int vertex_indices" in p.read_bytes()
        back = PlyData.read(str(p))["face"]
        assert [list(v) for v in back["vertex_indices"]] == [[0, 1, 2], [2, 3, 4, 5]] and list(back["flag"]) == [1, 0]
    with pytest.raises(plyfile.PlyParseError):
        PlyData.read(io.BytesIO(b"plx\n"))
    with pytest.raises(plyfile.PlyParseError):
        PlyData.read(io.BytesIO(b"ply\nformat binary_little_endian 1.0\nelement v 2\nproperty float x\nend_header\n\0\0"))
    with pytest.raises(TypeError):
        PlyElement.describe(np.zeros(3), "v")


# ---------------------------------------------------------------------------------- the reference's code on the shim
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")),
                               reason="reference tree only exists in the authoring container")


@pytest.fixture()
def ref_modules(monkeypatch):
    import torch
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    if not hasattr(np, "cast"):   # the reference calls np.cast[np.float16](..) (gaussian_model.py:269), removed in NumPy 2
        class _Cast:
            def __getitem__(self, dtype):
                return lambda a: np.asarray(a, dtype=dtype)
        monkeypatch.setattr(np, "cast", _Cast(), raising=False)
    for m in [k for k in sys.modules if k == "scene" or k.startswith("scene.") or k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, m)
    import scene.dataset_readers as dr
    import scene.gaussian_model as gm
    assert gm.PlyData is PlyData and dr.PlyElement is PlyElement      # the reference imported THIS module
    yield gm, dr
    for m in [k for k in sys.modules if k == "scene" or k.startswith("scene.") or k == "utils" or k.startswith("utils.")]:
        sys.modules.pop(m, None)


def _model(gm, P=1500, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    m = gm.GaussianModel(3)
    deg = torch.randint(0, 4, (P, 1), generator=g, dtype=torch.int32)
    m._degrees = deg
    m._xyz = torch.randn(P, 3, generator=g)
    m._features_dc = torch.randn(P, 1, 3, generator=g)
    rest = torch.randn(P, 15, 3, generator=g)
    keep = (torch.arange(15)[None, :] < ((deg + 1) ** 2 - 1)).unsqueeze(-1)   # culled bands are zero (cull_sh_bands)
    m._features_rest = rest * keep
    m._opacity = torch.randn(P, 1, generator=g)
    m._scaling = torch.randn(P, 3, generator=g)
    m._rotation = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    return m


def _by_degree(t, deg):
    import torch
    return torch.cat([t[(deg == d).squeeze()] for d in range(4)], dim=0)


@needs_ref
def test_reference_save_load_ply_plain(ref_modules, tmp_path):
    import torch
    gm, _ = ref_modules
    half = False   # the reference's loader only converts xyz back from half when not quantised (gaussian_model.py:363-364)
    m = _model(gm)
    path = str(tmp_path / "pc" / "point_cloud.ply")
    m.save_ply(str(tmp_path / "pc" / "half.ply"), quantised=False, half_float=True)
    hx = PlyData.read(str(tmp_path / "pc" / "half.ply")).elements[1]
    assert hx.ply_property("x").val_dtype == "i2" and hx.ply_property("opacity").val_dtype == "i2"
    assert np.array_equal(hx["x"].view(np.float16), m._xyz[(m._degrees == 1).squeeze(), 0].half().numpy())
    m.save_ply(path, quantised=False, half_float=half)
    raw = PlyData.read(path)
    assert [e.name for e in raw.elements] == ["vertex_0", "vertex_1", "vertex_2", "vertex_3"]
    assert [len(e.properties) for e in raw.elements] == [3 + 3 + 0 + 1 + 3 + 4, 14 + 9, 14 + 24, 14 + 45]
    assert raw.elements[0].ply_property("x").val_dtype == ("i2" if half else "f4")
    back = gm.GaussianModel(3)
    back.load_ply(path, half_float=half, quantised=False)
    deg = m._degrees
    assert torch.equal(back._degrees, _by_degree(deg, deg))

    def expect(t):
        t = _by_degree(t, deg)
        return t.half().float() if half else t
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(back, name).detach(), expect(getattr(m, name))), name


@needs_ref
@pytest.mark.parametrize("half", [False, True])
def test_reference_save_load_ply_quantised(ref_modules, tmp_path, half):
    import torch
    gm, _ = ref_modules
    m = _model(gm, P=900, seed=3)
    g = torch.Generator().manual_seed(5)
    P = 900

    def book(cols):
        return gm.Codebook(torch.randint(0, 256, (P, cols), generator=g, dtype=torch.uint8),
                           torch.randn(256, 1, generator=g))
    cb = OrderedDict()
    cb["features_dc"] = book(3)
    for i in range(15):
        cb[f"features_rest_{i}"] = book(3)
    cb["opacity"], cb["scaling"], cb["rotation_re"], cb["rotation_im"] = book(1), book(3), book(1), book(3)
    m._codebook_dict = cb
    path = str(tmp_path / "point_cloud_quantised.ply")
    m.save_ply(path, quantised=True, half_float=half)
    raw = PlyData.read(path)
    assert raw.elements[-1].name == "codebook_centers" and raw.elements[-1].count == 256
    assert [p.name for p in raw.elements[-1].properties] == list(cb.keys())
    assert raw.elements[2].ply_property("f_rest_3").val_dtype == "u1"
    back = gm.GaussianModel(3)
    back.load_ply(path, half_float=half, quantised=True)
    deg = m._degrees

    def centres(k):
        c = cb[k].centers.view(-1)
        return c.half().float() if half else c

    def lookup(k):
        return _by_degree(centres(k)[cb[k].ids.long()], deg)
    assert torch.equal(back._opacity.detach(), lookup("opacity"))
    assert torch.equal(back._scaling.detach(), lookup("scaling"))
    assert torch.equal(back._features_dc.detach(), lookup("features_dc").view(-1, 1, 3))
    assert torch.equal(back._rotation.detach(), torch.cat((lookup("rotation_re"), lookup("rotation_im")), dim=1))
    sd = _by_degree(deg, deg)
    rest = torch.stack([lookup(f"features_rest_{i}") for i in range(15)], dim=1)       # [P,15,3]
    # bands above a Gaussian's degree are not stored; the loader pads the INDEX with 0 -> centre 0 of that codebook
    pad = torch.stack([centres(f"features_rest_{i}")[0].expand(P, 3) for i in range(15)], dim=1)
    keep = (torch.arange(15)[None, :] < ((sd + 1) ** 2 - 1)).unsqueeze(-1)
    assert torch.equal(back._features_rest.detach(), torch.where(keep, rest, pad))


@needs_ref
def test_reference_store_and_fetch_point_cloud(ref_modules, tmp_path):
    _, dr = ref_modules
    rng = np.random.default_rng(2)
    xyz = rng.normal(0, 1, (500, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (500, 3)).astype(np.uint8)
    path = str(tmp_path / "points3D.ply")
    dr.storePly(path, xyz, rgb)
    pc = dr.fetchPly(path)
    assert np.array_equal(pc.points, xyz) and np.array_equal(pc.colors, rgb / 255.0) and not pc.normals.any()
