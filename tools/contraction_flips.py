"""How far does "bit-exact radii / tile IDs against the CUDA build" survive FMA contraction?  

nvcc contracts a * b + c into one FMA by default, so the reference's own integer outputs are defined only up to the
contraction pattern its compiler picks for forward.cu:189-201 (cov2D), :396-398 (projection), :429-432 (eigenvalues ->
radius) and auxiliary.h:46-56 (tile rect).  The oracle (and the product's preprocess kernel) are built with
-ffp-contract=off: one rounding per operation, reproducible everywhere.  This tool builds the oracle a SECOND time with
-ffp-contract=fast -mfma (oracle/Makefile: libraster_oracle_fma.so) and counts, per BASELINE.json config stand-in, how many
per-Gaussian integer outputs and how many entries of the sorted (tile << 32 | depth) list differ between the two builds.
gcc's contraction pattern is not nvcc's, so the numbers are a sensitivity estimate -- the honest error bar on a comparison
with a CUDA build -- not a prediction of which elements flip there.

    python tools/contraction_flips.py [workload ...]      (CPU only; ~1 min for the five stand-ins)

Each build runs in its own process (the oracle front-end loads one library per process).  Output: one line per workload,
also written to profiles/r06_contraction_flips.txt by the caller."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LISTS_UP_TO = 2_000_000   # the sorted list itself is compared up to this Gaussian count (memory / time)


CAM_SEED = 1   # a rotated, translated camera (bench.py's cameras 1..7 are such); the workloads' camera 0 is the identity pose,
               # under which most products of the projection are exact and contraction changes nothing


def run_one(name, out_path):
    import synth_scene as ss
    from oracle import oracle as orc
    w, cam0, g = ss.make_workload(name)     # the scene is placed in front of camera 0 ...
    W, H, P = w["W"], w["H"], w["P"]
    cam = ss.make_camera(W, H, w["f"], CAM_SEED)   # ... and seen from a camera with a general pose
    lists = P <= LISTS_UP_TO
    ref = orc.forward(np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0, None,
                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center, geometry_only=not lists)
    st = ref["state"]
    np.savez(out_path, radii=st["radii"], tiles=st["tiles_touched"], depth_bits=st["depths"].view(np.uint32), xy=st["xy"],
             num_rendered=np.int64(ref["num_rendered"]), keys=st["keys"] if lists else np.zeros(0, np.uint64),
             point_list=st["point_list"] if lists else np.zeros(0, np.uint32),
             n_contrib=st["n_contrib"] if lists else np.zeros(0, np.uint32))


def compare(name, a, b):
    P = a["radii"].size
    vis = (a["radii"] > 0) | (b["radii"] > 0)
    out = {"workload": name, "gaussians": int(P), "visible": int(vis.sum()),
           "num_rendered": [int(a["num_rendered"]), int(b["num_rendered"])],
           "radii_differ": int((a["radii"] != b["radii"]).sum()),
           "tiles_touched_differ": int((a["tiles"] != b["tiles"]).sum()),
           "depth_bits_differ": int(((a["depth_bits"] != b["depth_bits"]) & vis).sum()),
           "pixel_centre_bits_differ": int(((a["xy"].view(np.uint32) != b["xy"].view(np.uint32)).any(axis=1) & vis).sum())}
    if a["keys"].size and b["keys"].size:
        n = min(a["keys"].size, b["keys"].size)
        out["list_entries"] = [int(a["keys"].size), int(b["keys"].size)]
        out["list_keys_differ"] = int((a["keys"][:n] != b["keys"][:n]).sum()) + abs(int(a["keys"].size) - int(b["keys"].size))
        out["list_ids_differ"] = int((a["point_list"][:n] != b["point_list"][:n]).sum()) + abs(int(a["keys"].size) - int(b["keys"].size))
        # what matters to the image: the same (tile, Gaussian) SET in the same ORDER; depth bits that differ without
        # reordering anything change the exported key only
        out["pixels_n_contrib_differ"] = int((a["n_contrib"] != b["n_contrib"]).sum())
    return out


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--one":
        run_one(sys.argv[2], sys.argv[3])
        return
    names = sys.argv[1:] or ["cfg0_10k_400", "lego_like_300k_800", "metric_500k_1600x1062", "garden_like_2M_1600x1062",
                             "bicycle_like_5M_1600x1062", "train_like_6M_1920x1080"]
    for name in names:
        outs = {}
        for variant in ("", "fma"):
            path = f"/tmp/contraction_{name}_{variant or 'off'}.npz"
            env = dict(os.environ, R3_ORACLE_VARIANT=variant, OMP_NUM_THREADS=str(os.cpu_count() or 1))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", name, path], env=env)
            outs[variant] = np.load(path)
        print(json.dumps(compare(name, outs[""], outs["fma"])), flush=True)
        for variant in ("", "fma"):
            os.remove(f"/tmp/contraction_{name}_{variant or 'off'}.npz")


if __name__ == "__main__":
    main()
