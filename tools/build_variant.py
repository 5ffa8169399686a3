"""Builds libr3dgs_hip_<tag>.so with different compiler flags for ONE translation unit (A/B of codegen options):
    python tools/build_variant.py <tag> <unit.hip> [--drop FLAG]... -- <extra flags...>
Select it at run time with R3DGS_LIB=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))
import build as b  # noqa: E402

tag, unit = sys.argv[1], sys.argv[2]
rest = sys.argv[3:]
drop, extra = [], []
while rest and rest[0] == "--drop":
    drop.append(rest[1])
    rest = rest[2:]
if rest and rest[0] == "--":
    extra = rest[1:]
b.build(verbose=False)
objdir = os.path.join(b.HERE, "build")
flags = [f for f in b.COMMON + b.UNITS[unit] if f not in drop] + extra
obj = os.path.join(objdir, unit.replace(".hip", f"_{tag}.o"))
subprocess.check_call([b.HIPCC] + flags + ["-c", os.path.join(b.CSRC, unit), "-o", obj])
objs = [os.path.join(objdir, (u.replace(".hip", f"_{tag}.o") if u == unit else u.replace(".hip", ".o"))) for u in b.UNITS]
out = os.path.join(b.HERE, f"libr3dgs_hip_{tag}.so")
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
