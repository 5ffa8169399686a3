"""Static resources of every kernel of the library, from the compiler's own metadata (no GPU needed):
    python tools/kernel_resources.py > profiles/r05_kernel_resources.txt
Compiles each translation unit of reduced-3dgs_amd/build.py to gfx950 assembly with the build's flags and reads the
.amdgpu_metadata notes: VGPRs / AGPRs / SGPRs, LDS bytes, scratch bytes (spills), workgroup size, and from those the waves per SIMD
a kernel can hold (512 VGPRs per SIMD lane on gfx950, 160 KB of LDS per CU)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))
import build as b  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


rows = []
with tempfile.TemporaryDirectory() as tmp:
    for unit, extra in b.UNITS.items():
        asm = os.path.join(tmp, unit.replace(".hip", ".s"))
        flags = [f for f in b.COMMON + extra if f != "-fPIC"]
        subprocess.run([b.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", asm, os.path.join(b.CSRC, unit)],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
        meta = txt[txt.index(".amdgpu_metadata"):] if ".amdgpu_metadata" in txt else ""
        for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
            blk = ".agpr_count:" + blk
            get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))  # noqa: E731
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            rows.append(dict(unit=unit, name=name, vgpr=get("vgpr_count"), agpr=get("agpr_count"), sgpr=get("sgpr_count"),
                             lds=get("group_segment_fixed_size"), scratch=get("private_segment_fixed_size"),
                             wg=get("max_flat_workgroup_size")))
for r, d in zip(rows, demangle([r["name"] for r in rows])):
    d = d.replace("void ", "").replace("(anonymous namespace)::", "").replace("r3::", "")
    r["name"] = re.sub(r"\(.*", "", d)
print("kernel resources on gfx950 (compiler metadata of this tree's build; waves/SIMD = what registers and STATIC LDS allow, 8 at most;")
print("the depth_sort_color / depth_colscan kernels also take dynamic LDS -- up to 64 KB per workgroup, capi.hip prepare_depth_bucket_sort --")
print("which this table does not see; scratch = spilled registers, all outside the kernels' inner loops)")
print(f"{'kernel':62s} {'unit':20s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS B':>7s} {'scratch B':>9s} {'wg':>5s} {'waves/SIMD':>10s}")
rows = [r for r in rows if "rocprim" not in r["name"] and "hipcub" not in r["name"]]   # the library's own kernels
for r in sorted(rows, key=lambda r: (r["unit"], r["name"])):
    regs = max(r["vgpr"] + r["agpr"], 1)
    granule = (regs + 7) // 8 * 8
    by_regs = min(8, 512 // granule)
    waves_per_wg = (r["wg"] + 63) // 64
    by_lds = 8 if r["lds"] == 0 else min(8, (160 * 1024 // r["lds"]) * waves_per_wg // 4)
    print(f"{r['name'][:62]:62s} {r['unit']:20s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['lds']:7d} {r['scratch']:9d} {r['wg']:5d} "
          f"{min(by_regs, max(by_lds, 0)):10d}")
