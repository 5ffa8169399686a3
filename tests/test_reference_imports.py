"""The reference's unmodified entry points import against THIS repository's drop-in packages
(`diff_gaussian_rasterization`, `simple_knn`, `plyfile` under reduced-3dgs_amd/) -- the "switch PYTHONPATH and go"
claim of INTEGRATION.md.  Authoring container only: the reference tree is not present on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")), reason="reference tree only exists in the authoring container")
def test_train_scene_renderer_compress_import_on_our_packages(tmp_path):
    code = (
        "import importlib, os, sys\n"
        "for m in ('scene', 'gaussian_renderer', 'train', 'compress'):\n"
        "    importlib.import_module(m)\n"
        "import diff_gaussian_rasterization, simple_knn._C, plyfile\n"
        "pkg = %r\n"
        "for mod in (diff_gaussian_rasterization, simple_knn._C, plyfile):\n"
        "    assert os.path.abspath(mod.__file__).startswith(pkg), mod.__file__\n"
        "print('imports-ok')\n") % os.path.join(ROOT, "reduced-3dgs_amd")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "reduced-3dgs_amd"), REF]))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), capture_output=True, text=True,
                         timeout=300)
    assert "imports-ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
