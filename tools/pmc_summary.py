"""Per-kernel means of the rocprofv3 --pmc passes written by tools/gpu_round.sh (PMC="...;...") into
gpurun_out/pmc*/..._counter_collection.csv  ->  one JSON {kernel: {counter: mean per launch}}.

    python tools/pmc_summary.py [out.json [dir-prefix]]   (default: gpurun_out/pmc_summary.json from gpurun_out/pmc*;
                                                          a prefix such as pmc2M_ takes gpurun_out/pmc2M_*)

Only kernels of this library (r3::*) are kept; template arguments stay in the name, parameter lists are cut.
Every pass is a separate run of `bench.py --steps 3 --warmup 1`, so the launch counts of the passes agree."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:           # cut the parameter list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace("(anonymous namespace)::", "")


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_summary.json")
    prefix = sys.argv[2] if len(sys.argv) > 2 else "pmc"
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dirs = [d for d in glob.glob(os.path.join(ROOT, "gpurun_out", prefix + "*")) if os.path.isdir(d) and
            (prefix != "pmc" or os.path.basename(d)[3:].isdigit())]   # plain "pmc": pmc0, pmc1 ... only
    for path in sorted(p_ for d in dirs for p_ in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = defaultdict(float)   # a counter is reported once per XCD/SE instance: sum them per dispatch
        names = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if not k.startswith("r3::"):
                    continue
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = k
        for (disp, ctr), v in per_dispatch.items():
            a = acc[names[disp]][ctr]
            a[0] += v
            a[1] += 1
    summary = {k: {c: round(v[0] / v[1], 1) for c, v in sorted(ctrs.items())} for k, ctrs in sorted(acc.items())}
    sys.path.insert(0, ROOT)
    import bench   # the fingerprint of the kernel sources these counters were collected on (bench.py compares it)
    summary["_meta"] = {"sources_sha16": bench.kernel_sources_sha16()}
    json.dump(summary, open(out_path, "w"), indent=1)
    for k, ctrs in summary.items():
        if k.startswith("_"):
            continue
        print(k, {c: ctrs[c] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU") if c in ctrs})


if __name__ == "__main__":
    main()
