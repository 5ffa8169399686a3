"""Small seeded end-to-end cases shared by tests/golden/make_golden.py (which stores the
oracle's outputs for them) and the parity tests (which rebuild the same inputs)."""
import numpy as np

import synth_scene as ss

CASES = {
    "a_deg3_black": dict(P=600, W=72, H=50, f=60.0, cam_seed=None, gseed=0, degree_mode="all3", bg=(0, 0, 0),
                         lam=0.0),
    "b_mixed_white_sparsity": dict(P=600, W=72, H=50, f=60.0, cam_seed=2, gseed=1, degree_mode="mixed",
                                   bg=(1, 1, 1), lam=0.1),
    "c_deg0_rand": dict(P=900, W=64, H=64, f=50.0, cam_seed=4, gseed=2, degree_mode="all0", bg=(0.2, 0.7, 0.4),
                        lam=0.0),
}


def case_inputs(kw):
    cam = ss.make_camera(kw["W"], kw["H"], kw["f"], kw["cam_seed"])
    g = ss.make_gaussians(kw["P"], cam, seed=kw["gseed"], degree_mode=kw["degree_mode"], scale_mu=0.12,
                          scale_sigma=0.7)
    bg = np.array(kw["bg"], np.float32)
    dl = ss.upstream_grad(kw["W"], kw["H"], seed=kw["gseed"] + 10) * (kw["W"] * kw["H"])
    return cam, g, bg, dl
