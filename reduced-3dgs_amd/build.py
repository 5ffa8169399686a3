"""Builds libr3dgs_hip.so (the C-ABI rasterizer library) for gfx950 with hipcc, in-tree.

    python reduced-3dgs_amd/build.py [--force]

Per-translation-unit flags matter for parity:
  * preprocess*.hip : -ffp-contract=off + correctly rounded fp32 divide/sqrt, so the integer outputs
                      (radii, tile rects, tiles_touched, sort order) are reproducible bit-for-bit;
  * blend.hip       : contraction allowed (FMA) + hardware exp: compared to 1e-5 / 1e-4, not bitwise;
                      -fno-slp-vectorize: SLP packing into v_pk_*_f32 cost 22 VGPRs (130 -> 108, one more
                      wave per SIMD), ~40 v_mov shuffles and blocked the v_add_f32_dpp fusion of the reductions;
  * -munsafe-fp-atomics: float atomicAdd lowers to global_atomic_add_f32 instead of a CAS loop.
hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libr3dgs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
          "-Wno-unused-function", "-I", CSRC] + (
    # the first kernel arguments (the pointer into the pass block) arrive in SGPRs instead of through a load
    [] if os.environ.get("R3DGS_NO_KERNARG_PRELOAD") else ["-mllvm", "-amdgpu-kernarg-preload-count=8"])
EXACT = ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]
UNITS = {  # depth_sort.h roles are instantiated in preprocess.hip (fused with the colour stream)
    "preprocess.hip": EXACT,
    "preprocess_bwd.hip": EXACT,
    "binning.hip": [],
    "colour_variance.hip": EXACT,
    "reduction_ops.hip": EXACT,
    "knn.hip": EXACT,
    "kmeans.hip": EXACT,
    # max-ilp scheduling: blend_bwd 0.497 -> 0.484 ms, blend_fwd 0.179 -> 0.176 (max-memory-clause: no change; -O2: worse;
    # with SLP vectorisation: 0.76 / 0.20)
    "blend.hip": ["-ffp-contract=fast", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "capi.hip": [],
}
HEADERS = ["common.h", "gauss_math.h", "blend_math.h", "depth_sort.h", os.path.join("..", "..", "include", "r3dgs_rasterizer.h"),
           os.path.join("..", "..", "include", "r3dgs_reduction.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    global OUT
    # whole-library A/B builds: R3DGS_BUILD_TAG=<tag> R3DGS_EXTRA_FLAGS="-D..." -> libr3dgs_hip_<tag>.so (own object directory)
    tag, extra_all = os.environ.get("R3DGS_BUILD_TAG"), os.environ.get("R3DGS_EXTRA_FLAGS", "").split()
    if tag:
        OUT = os.path.join(HERE, f"libr3dgs_hip_{tag}.so")
    objdir = os.path.join(HERE, "build" + (f"_{tag}" if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    hdr_time = _newest([os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)])
    jobs = []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append(([HIPCC] + COMMON + extra + extra_all + ["-c", s, "-o", o], src))
    def run(job):
        cmd, name = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stdout + r.stderr
    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for name, rc, log in ex.map(run, jobs):
            if verbose and (rc != 0 or log.strip()):
                print(f"--- {name} (rc={rc})\n{log}")
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in UNITS]
    if jobs or force or not os.path.exists(OUT):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    if not tag:
        build_torch_binding(force=force, verbose=verbose)
    return OUT


TORCH_EXT_SRC = os.path.join(HERE, "csrc_torch", "r3dgs_torch.cpp")
TORCH_EXT_OUT = os.path.join(HERE, "diff_gaussian_rasterization", "_r3dgs_torch.so")


def build_torch_binding(force=False, verbose=True):
    """The compiled torch binding of the hot calls (csrc_torch/r3dgs_torch.cpp -> diff_gaussian_rasterization/
    _r3dgs_torch.so): a pybind module compiled with the host compiler against torch's own headers -- it contains no device
    code and does not link against libr3dgs_hip.so (it is handed the library's entry points at import), so no hipify pass
    and no hipcc are involved.  ~30 s."""
    import sysconfig

    import torch
    hdr = os.path.join(HERE, "..", "include", "r3dgs_rasterizer.h")
    if not force and os.path.exists(TORCH_EXT_OUT) and os.path.getmtime(TORCH_EXT_OUT) >= max(
            os.path.getmtime(TORCH_EXT_SRC), os.path.getmtime(hdr), os.path.getmtime(torch.__file__)):
        return TORCH_EXT_OUT
    tdir = os.path.dirname(torch.__file__)
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", TORCH_EXT_OUT, TORCH_EXT_SRC,
           "-I", os.path.join(HERE, "..", "include"), "-I", os.path.join(tdir, "include"),
           "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", sysconfig.get_paths()["include"],
           "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_r3dgs_torch",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-L", os.path.join(tdir, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
           "-Wl,-rpath," + os.path.join(tdir, "lib"), "-Wno-deprecated-declarations"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if verbose:
            print(r.stdout + r.stderr)
        raise RuntimeError("building the torch binding failed")
    return TORCH_EXT_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
