"""Copies the outputs of tools/refresh_profiles.sh (gpurun_out/) into profiles/r06_* and regenerates profiles/README.md."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
R = "r06"


def last_line(src, dst, mode="w"):
    lines = [l for l in open(os.path.join(G, src)) if l.startswith("{")]
    open(os.path.join(P, dst), mode).write(lines[-1])


last_line("bench.log", f"{R}_bench_n1_steps20.json")
last_line("bench_default.log", f"{R}_bench_n1_default.json")
last_line("bench_nonstrict.log", f"{R}_bench_n1_ab.jsonl")
last_line("bench_refrects.log", f"{R}_bench_n1_ab.jsonl", "a")
last_line("bench_burner64_a.log", f"{R}_bench_n1_cpu_burner64.jsonl")
last_line("bench_burner64_b.log", f"{R}_bench_n1_cpu_burner64.jsonl", "a")
last_line("bench_noseg.log", f"{R}_bench_n1_ab.jsonl", "a")
last_line("bench_f32chain.log", f"{R}_bench_n1_ab.jsonl", "a")
shutil.copy(os.path.join(G, "prof", "r_kernel_stats.csv"), os.path.join(P, f"{R}_kernel_stats_bench_500k_1600x1062.csv"))
shutil.copy(os.path.join(G, "pmc_summary.json"), os.path.join(P, f"{R}_pmc_summary.json"))
for tag in ("2M", "6M", "clustered_500k"):
    if os.path.exists(os.path.join(G, f"pmc_summary_{tag}.json")):
        shutil.copy(os.path.join(G, f"pmc_summary_{tag}.json"), os.path.join(P, f"{R}_pmc_summary_{tag}.json"))
if os.path.exists(os.path.join(G, "valu_rate.txt")):
    shutil.copy(os.path.join(G, "valu_rate.txt"), os.path.join(P, f"{R}_valu_rate.txt"))
for extra in ("sweep_colour_split.txt", "dryrun_2rank_line.json"):
    if os.path.exists(os.path.join(G, extra)):
        shutil.copy(os.path.join(G, extra), os.path.join(P, f"{R}_{extra}"))
if os.path.exists(os.path.join(G, "other_workloads.jsonl")):
    shutil.copy(os.path.join(G, "other_workloads.jsonl"), os.path.join(P, f"{R}_other_workloads.jsonl"))
for tl in ("bwd_timeline", "bwd_timeline_clustered", "bwd_timeline_clustered_whole_lists", "fwd_timeline"):
    if os.path.exists(os.path.join(G, tl + ".txt")):
        shutil.copy(os.path.join(G, tl + ".txt"), os.path.join(P, f"{R}_{tl}.txt"))
open(os.path.join(P, f"{R}_gpu_tests.txt"), "w").write(open(os.path.join(G, "pytest_gpu.log")).read() +
                                                       "\n" + open(os.path.join(G, "smoke.log")).read()[-1600:])
if os.path.exists(os.path.join(G, "ab_rounds.txt")):
    shutil.copy(os.path.join(G, "ab_rounds.txt"), os.path.join(P, f"{R}_ab_round5_vs_round6.txt"))
if os.path.exists(os.path.join(G, "host_bound.log")):
    shutil.copy(os.path.join(G, "host_bound.log"), os.path.join(P, f"{R}_host_bound_bindings.txt"))
for wl, short in (("garden_like_2M_1600x1062", "garden_like_2M"), ("train_like_6M_1920x1080", "train_like_6M"),
                  ("clustered_500k_1600x1062", "clustered_500k"), ("garden_clustered_2M", "garden_clustered_2M")):
    f = os.path.join(G, f"prof_{wl}", "r_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, f"{R}_kernel_stats_{short}.csv"))
if "--no-readme" not in sys.argv:
    sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "profiles_readme.py")]))
