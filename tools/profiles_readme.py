"""Regenerates profiles/README.md from the committed measurement files of the round (profiles/r06_*).

    python tools/profiles_readme.py

Numbers in that README are therefore exactly the ones in the JSON / CSV / txt files next to it."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = os.path.join(ROOT, "profiles")
R, PREV = "r06", "r05"   # bracketed numbers: the previous round's files


def jl(name):
    return [json.loads(l) for l in open(os.path.join(P, name)) if l.strip()]


def kstats(name):
    ks = {}
    for r in csv.DictReader(open(os.path.join(P, name))):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        ks[n.split("(")[0]] = float(r["AverageNs"]) / 1e3
    return ks


import bench  # noqa: E402  (traffic and the calibrated VALU floor are computed by bench.py's own functions)

d20 = jl(f"{R}_bench_n1_steps20.json")[0]
d50 = jl(f"{R}_bench_n1_default.json")[0]
ab = {k: v for k, v in zip(("nonstrict", "refrects", "whole_lists", "f32chain"), jl(f"{R}_bench_n1_ab.jsonl"))}
burn = jl(f"{R}_bench_n1_cpu_burner64.jsonl")
others = {o["config"]["workload"]: o for o in jl(f"{R}_other_workloads.jsonl")}
old_others = {o["config"]["workload"]: o for o in jl(f"{PREV}_other_workloads.jsonl")}
old50 = jl(f"{PREV}_bench_n1_default.json")[0]
ks = kstats(f"{R}_kernel_stats_bench_500k_1600x1062.csv")
ks_old = kstats(f"{PREV}_kernel_stats_bench_500k_1600x1062.csv")
st = d50["stages"]
STAGES = ("preprocess_fwd", "depth_sort_scan", "tile_binning", "blend_fwd", "blend_bwd", "preprocess_bwd")


def stages_of(d):
    return " / ".join(f"{d['stages'][k]['avg_ms']:.3f}" for k in STAGES)


def kern(table, prefix):
    v = bench.find_kernel(table, prefix)
    return float("nan") if v is None else v


KERNELS = [("preprocess_geom_kernel<0>", "r3::preprocess_geom_kernel<0>"), ("depth_sort_color<0> (histogram + header + colour)", "r3::depth_sort_color_kernel<0, false>"),
           ("depth_colscan", "r3::depth_colscan_kernel"), ("depth_sort_color<1> (scatter + colour)", "r3::depth_sort_color_kernel<1, false>"),
           ("depth_sort_color<2> (bucket sort + scan + colour)", "r3::depth_sort_color_kernel<2, false>"),
           ("emit_pairs", "r3::emit_pairs_kernel<"), ("radix_digit_scan (x2)", "r3::radix_digit_scan_kernel"),
           ("radix_scatter (x2)", "r3::radix_scatter_kernel<"), ("radix_hist", "r3::radix_hist_kernel<"),
           ("tile_ranges", "r3::tile_ranges_kernel<"), ("blend_fwd_kernel<1>", "r3::blend_fwd_kernel<1, false>"),
           ("unit_order_kernel [tile_order_kernel]", "r3::unit_order_kernel"), ("blend_bwd_kernel<4, true, false>", "r3::blend_bwd_kernel<4, true, false>"),
           ("pair_reduce", "r3::pair_reduce_kernel"), ("preprocess_bwd_kernel<true, true>", "r3::preprocess_bwd_kernel<")]
WL = [("metric_500k_1600x1062", "bench_500k_1600x1062", ""), ("clustered_500k_1600x1062", "clustered_500k", "_clustered_500k"),
      ("garden_like_2M_1600x1062", "garden_like_2M", "_2M"), ("train_like_6M_1920x1080", "train_like_6M", "_6M")]


def kernel_table():
    cols = {w: kstats(f"{R}_kernel_stats_{short}.csv") for w, short, _ in WL}
    old = {"metric_500k_1600x1062": ks_old, "garden_like_2M_1600x1062": kstats(f"{PREV}_kernel_stats_garden_like_2M.csv"),
           "train_like_6M_1920x1080": kstats(f"{PREV}_kernel_stats_train_like_6M.csv")}
    out = ["| kernel (µs, rocprofv3 --kernel-trace --stats; [round 5]) | " + " | ".join(w for w, _, _ in WL) + " |", "|---|" + "---|" * len(WL)]
    for label, prefix in KERNELS:
        cells = []
        for w, _, _ in WL:
            c = f"{kern(cols[w], prefix):.1f}"
            if w in old:
                c += f" [{kern(old[w], prefix):.1f}]"
            cells.append(c)
        out.append(f"| `{label}` | " + " | ".join(cells) + " |")
    return "\n".join(out)


def traffic_table():
    out = ["| stage: counter bytes per launch (MB; FETCH_SIZE x2 on the wide-load kernels + WRITE_SIZE), its time in this bench line, GB/s of bytes that moved | " +
           " | ".join(w for w, _, _ in WL) + " |", "|---|" + "---|" * len(WL)]
    lines = {"metric_500k_1600x1062": d50, **{w: others[w] for w, _, _ in WL[1:] if w in others}}
    for s_ in STAGES:
        cells = []
        for w, _, _ in WL:
            d = lines.get(w)
            if d is None or d["stages"][s_].get("counter_bytes") is None:
                cells.append("–")
            else:
                x = d["stages"][s_]
                cells.append(f"{x['counter_bytes'] / 1e6:.0f} MB, {x['avg_ms']:.3f} ms, **{x['counter_GBps']:.0f}**")
        out.append(f"| {s_} | " + " | ".join(cells) + " |")
    cells = []
    for w, _, _ in WL:
        d = lines.get(w)
        ir = d["iter_roofline"] if d else None
        cells.append("–" if not ir or ir["frac_counter_traffic"] is None else
                     f"{ir['counter_traffic_bytes'] / 1e9:.2f} GB/step = **{100 * ir['frac_counter_traffic']:.1f} %** of 8 TB/s moved; "
                     f"{100 * ir['frac_of_8TBps']:.1f} % in SURVEY §8d's reference-algorithm bytes")
    out.append("| whole iteration | " + " | ".join(cells) + " |")
    return "\n".join(out)


def others_table():
    out = ["| workload | it/s [round 5] | ms/step | stages ms: " + " / ".join(STAGES) + " | render FPS |", "|---|---|---|---|---|"]
    for w, o in others.items():
        prev = f" [{old_others[w]['value']}]" if w in old_others else ""
        out.append(f"| {w} (P = {o['config']['gaussians']}, R̄ = {o['config']['num_rendered_mean'] / 1e6:.2f} M, binned "
                   f"{o['config']['pairs_binned_mean'] / 1e6:.2f} M) | **{o['value']}**{prev} | {o['ms_per_step']} | {stages_of(o)}"
                   + (f" (+ sh_color {o['stages']['sh_color']['avg_ms']:.3f}: generic depth sort)" if "sh_color" in o["stages"] else "")
                   + f" | {o['render_fps']} |")
    return "\n".join(out)


def txt(name, skip="/opt"):
    p_ = os.path.join(P, name)
    if not os.path.exists(p_) and name.startswith(R + "_"):   # not re-collected this round: the previous round's file
        p_ = os.path.join(P, PREV + name[len(R):])
    return "".join(l for l in open(p_) if not l.startswith(skip)) if os.path.exists(p_) else "(not collected)\n"


vb = bench.pmc_valu("blend_bwd", "metric_500k_1600x1062", st["blend_bwd"]["avg_ms"])
vf = bench.pmc_valu("blend_fwd", "metric_500k_1600x1062", st["blend_fwd"]["avg_ms"])
VR = os.path.join(P, f"{R}_valu_rate.txt") if os.path.exists(os.path.join(P, f"{R}_valu_rate.txt")) else os.path.join(P, f"{PREV}_valu_rate.txt")
f64 = [l.rstrip() for l in open(VR) if "f64" in l and " 8 waves/SIMD" in l]
f32 = [l.rstrip() for l in open(VR) if l.startswith("v_fma_f32 ") and " 8 waves/SIMD" in l]

new = f'''# profiles/ — measurements on MI355X (round 6)

Every `{R}_*` file comes from visits of `tools/refresh_profiles.sh` to a 1-GPU MI355X box (gfx950, ROCm 7.2, torch 2.10+rocm7.0;
256 visible CPUs, **cgroup CPU quota 16**); `tools/collect_profiles.py` copies the results here and runs this generator
(`tools/profiles_readme.py`), so the numbers below are the ones in the files.  The command profiled is `python bench.py`
(workload `metric_500k_1600x1062`: 500 000 Gaussians, 1600×1062, SH degree 3, SURVEY.md §8d recipe, 8 cameras; V̄ =
{d50['config']['visible_mean'] / 1e3:.0f}k visible, `num_rendered` R̄ = {d50['config']['num_rendered_mean'] / 1e6:.2f} M — the reference's count, which the byte formulas are
written in — of which {d50['config']['pairs_binned_mean'] / 1e6:.2f} M (tile, Gaussian) pairs are binned, N = 1.70 Mpix) unless a workload is named.  These are
this build's own visits; **the number of record is the driver's `BENCH_r06.json`**.  Boxes differ: one visit of this round measured
every stage 1.2–1.4× slower than the others (1057 it/s for the same build), so every A/B below is from ONE visit, with the
previous round's library (`R3DGS_LIB=old`) run beside the new one where rounds are compared.

| file | what |
|---|---|
| `{R}_bench_n1_default.json`, `{R}_bench_n1_steps20.json` | the JSON line of `python bench.py` (50 steps, 10 warm-up) and of the driver's form (`--steps 20 --warmup 5`) |
| `{R}_bench_n1_ab.jsonl` | the driver's form with `R3DGS_STRICT=0`, `R3DGS_TIGHT_RECT=0` (the reference's 3σ squares), `R3DGS_BWD_SEG=0` (whole tile lists: no list segments) and `R3DGS_F64_CHAIN=0` (the reference's fp32 covariance chain), in this order |
| `{R}_bench_n1_cpu_burner64.jsonl` | the driver's form twice while `tools/cpu_burn.py 64` exhausts the container's CPU quota |
| `{R}_kernel_stats_*.csv` | `rocprofv3 --kernel-trace --stats` of `bench.py --steps 10 --warmup 3` on the metric workload, the clustered 500 k scene, 2 M, 6 M @1920×1080 and the clustered 2 M scene |
| `{R}_pmc_summary.json` | (`tools/pmc_summary.py`) per-kernel means of four separate `rocprofv3 --kernel-trace --pmc …` passes (VALU instructions by class; SQ activity / wait / LDS counters; `FETCH_SIZE`; `WRITE_SIZE`) of `bench.py --steps 3 --warmup 1`; `bench.py` cites it as `roofline.traffic`, `roofline.valu`, `stages.*.counter_bytes` |
| `{R}_pmc_summary_2M.json`, `_6M.json`, `_clustered_500k.json` | the `FETCH_SIZE` / `WRITE_SIZE` passes of the same command on those workloads (`bench.py --workload …` cites them) |
| `{R}_ab_round5_vs_round6.txt` | five workloads through round 5's library and this round's, alternating, in one visit |
| `{R}_other_workloads.jsonl` | `tools/other_workloads.sh`: bench.py lines of the configs[0..4] stand-ins and of the two clustered scenes |
| `{R}_bwd_timeline.txt`, `{R}_bwd_timeline_clustered.txt`, `{R}_bwd_timeline_clustered_whole_lists.txt` | `tools/bwd_timeline.py` (debug build `-DR3_TIMELINE`): when and where every workgroup of the backward blend ran — metric scene, clustered scene with list segments (default) and without |
| `{R}_bwd_tail_model_clustered_500k.txt` | `tools/bwd_tail_model.py clustered_500k_1600x1062`: tile-weight histogram of the clustered scene and the processor-sharing model of its schedule |
| `{R}_lane_utilisation.txt` | `tools/lane_utilisation.py` (CPU, the kernels' own per-lane functions): useful lanes per evaluated (entry, quadrant) pair of both blend kernels, and what a 4×4-granular pre-test with per-row lists would evaluate |
| `{R}_kernel_resources.txt` | `tools/kernel_resources.py` (no GPU: the compiler's metadata for this tree's build): registers, static LDS, scratch and the waves per SIMD they allow, for every kernel of the library |
| `{R}_exp_list_blocks.txt`, `{R}_exp_list_blocks2.txt` | `tools/exp_list_blocks.sh`, `tools/exp_list_blocks2.sh` (the follow-up on the final tree: 2×2 / 8×8 against the build's 4×4): the backward's unit lists as every eighth tile / 2×2 / 4×4 tile blocks, three builds alternating in one visit, with the blend kernels' `FETCH_SIZE` / `WRITE_SIZE` (4×4 is the build) |
| `{R}_exp_unit_order.txt` | `tools/exp_unit_order.sh`: the unit order kernel without its first pass and with two instead of eight register slots per thread (variant builds, not in the tree): 12.7 → 11.4 → 7.5 µs |
| `{R}_exp_pair_reduce_latency.txt`, `{R}_sweep_depth_bucket_load.txt` | one-visit A/Bs that changed nothing and are recorded as such: `pair_reduce` with its loads hoisted (one memory round trip instead of three), depth-sort bucket loads of 64 … 512 |
| `{R}_valu_rate.txt` | `tools/valu_rate.hip`: cycles per wave64 instruction and SIMD for 25 instruction kinds at 1 / 2 / 4 / 5 / 8 waves per SIMD, now with `v_fma_f64 / v_mul_f64 / v_add_f64` |
| `{R}_gpu_tests.txt` | `pytest tests -m gpu -s`: every gradient distance measured (HIP vs the fp32 oracle, HIP vs the double evaluation, fp32 oracle vs the double evaluation, the fp32-chain mode, segments vs whole lists), the own-loop report |
| `{R}_host_bound_bindings.txt` | `tools/host_bound_bench.py`: steps/s of the 10k and 300k workloads through the ctypes marshalling and the compiled torch binding |
| `{R}_contraction_flips.txt` | `tools/contraction_flips.py` (CPU): integer outputs that differ between the oracle built with `-ffp-contract=off` and with `-ffp-contract=fast -mfma`, per BASELINE config stand-in, seen from a rotated camera — the error bar on "bit-exact vs a CUDA build" (INTEGRATION.md §5) |
| `{R}_exp_prebwd_lean.txt` | one-visit A/B of a lean form of the per-Gaussian backward (half-wave LDS windows, 3 / 4 / 5 waves per SIMD by launch bounds): no gain at three waves, 20 % / 85 % slower with the spills the fourth / fifth wave cost — not in the tree |
| `{R}_sweep_colour_split.txt` | `tools/ab.sh`: colour-stream grid / split and depth-bucket-load knobs around the defaults, two alternating rounds |
| `{R}_dryrun_2rank_line.json` | `tools/dryrun_2rank.py`: the bench line of the N > 1 code path run as two ranks on ONE GPU over gloo (functional only): `exchange_forms` lists every compact transport with its bytes and the model's prediction |
| `r01_*` … `r05_*` | earlier rounds, kept for comparison |

## Headline (N = 1)

| | value |
|---|---|
| training iterations/s (fwd + bwd through the autograd boundary, 1 view/iter, strict), 50 steps | **{d50['value']} it/s** ({d50['ms_per_step']} ms/step; GPU first-to-last kernel {d50['host']['gpu_event_ms_per_step']} ms/step; Σ stage {d50['host']['gpu_stage_ms_sum']} ms) [round 5: {old50['value']}] |
| the driver's form, 20 steps / 5 warm-up | **{d20['value']} it/s** ({d20['ms_per_step']} ms/step = {d20['host']['step_over_gpu_stage_sum']} × Σ stage ms); passes in the timed region: {d20['config']['passes_in_timed_region']} |
| stages ms ({' / '.join(STAGES)}) | {stages_of(d50)} [round 5: {stages_of(old50)}] |
| same, strict mode off | {ab['nonstrict']['value']} it/s |
| `lambda_sh_sparsity = 0.1` (full_eval.py:33,44) | {d50['value_sh_sparsity']} it/s ({d50['sh_sparsity']['ms_per_step']} ms/step) |
| reference rects (`R3DGS_TIGHT_RECT=0`) | {ab['refrects']['value']} it/s, stages {stages_of(ab['refrects'])} |
| whole tile lists (`R3DGS_BWD_SEG=0`) | {ab['whole_lists']['value']} it/s, `blend_bwd` stage {ab['whole_lists']['stages']['blend_bwd']['avg_ms']} ms against {d20['stages']['blend_bwd']['avg_ms']} |
| fp32 covariance chain (`R3DGS_F64_CHAIN=0`) | {ab['f32chain']['value']} it/s, `preprocess_bwd` {ab['f32chain']['stages']['preprocess_bwd']['avg_ms']} ms against {d20['stages']['preprocess_bwd']['avg_ms']} |
| under `cpu_burn 64` | {burn[0]['value']} / {burn[1]['value']} it/s |
| render only (`render.py`'s FPS path) | {d50['render_fps']} FPS = {d50['render_mpix_per_s']} Mpix/s |
| dominant kernel against the HBM roof (contract form) | `blend_bwd` stage: {d50['roofline']['achieved']} GB/s of SURVEY §8d bytes = **{d50['roofline']['frac']}** of 8 TB/s; bytes that moved per launch (counters): {d50['roofline']['traffic'] / 1e6:.0f} MB |
| dominant kernel against the roof that binds it (VALU issue) | `blend_bwd_kernel`: {vb['insts'] / 1e6:.1f} M VALU instructions, floor {vb['floor_ms']} ms / kernel {vb['kernel_ms_committed_profile']} ms = **{vb['floor_ms'] / vb['kernel_ms_committed_profile']:.2f}** [round 5: 0.84]; `blend_fwd_kernel`: floor {vf['floor_ms']} / {vf['kernel_ms_committed_profile']} ms = {vf['floor_ms'] / vf['kernel_ms_committed_profile']:.2f} |
| whole iteration | `B_iter` = {d50['iter_roofline']['B_iter_bytes'] / 1e9:.3f} GB (reference-algorithm bytes) → {100 * d50['iter_roofline']['frac_of_8TBps']:.1f} % of 8 TB/s (target 40 %); bytes the counters saw move: {d50['iter_roofline']['counter_traffic_bytes'] / 1e9:.2f} GB/step = {100 * d50['iter_roofline']['frac_counter_traffic']:.1f} % |
| CPU baseline (`cpu_baseline`, kind "port") | {d50['cpu_baseline']['value']} it/s on {d50['cpu_baseline']['cores']} threads of a {d50['cpu_baseline'].get('cpu_model')}: {d50['cpu_baseline']['sample'][:110]}…; the bench workload itself by the C restatement: {d50['cpu_baseline']['same_workload_sample']['value']} it/s |

## This round's library against round 5's, same visit (`{R}_ab_round5_vs_round6.txt`: `bench.py --steps 20 --warmup 5 --cameras 4`, `R3DGS_LIB=old` = the round-5 tree built beside)

```
{txt(f"{R}_ab_round5_vs_round6.txt")}```

## Kernels

{kernel_table()}

## HBM traffic from the counters, per stage and workload

{traffic_table()}

`alg_GBps` in the bench lines is SURVEY §8d's reference-algorithm bytes over the stage time — it counts bytes this library never
moves (the reference's 64-bit key sort) and can exceed the HBM peak; `counter_GBps` above is what moved.

## Other workloads

{others_table()}

## The backward blend on a scene with a real capture's load (clustered_500k_1600x1062)

Tile weights (`{R}_bwd_tail_model_clustered_500k.txt`, first lines):

```
{"".join(open(os.path.join(P, f"{R if os.path.exists(os.path.join(P, f'{R}_bwd_tail_model_clustered_500k.txt')) else PREV}_bwd_tail_model_clustered_500k.txt")).readlines()[:3])}```

Whole tile lists (`R3DGS_BWD_SEG=0`; one workgroup per tile, heaviest first) — the chip drains for the last third of the kernel:

```
{txt(f"{R}_bwd_timeline_clustered_whole_lists.txt")}```

List segments (default): lists of at least max(256, 0.75 × the pass's mean list length) entries are walked in segments of 128 by
several workgroups, from checkpoints the forward leaves:

```
{txt(f"{R}_bwd_timeline_clustered.txt")}```

The metric scene with list segments:

```
{txt(f"{R}_bwd_timeline.txt")}```

## Lane utilisation of the blend kernels (CPU, the kernels' own per-lane functions)

```
{txt(f"{R}_lane_utilisation.txt")}```

## fp64 on the vector unit (`{R}_valu_rate.txt`, 8 waves per SIMD)

```
{chr(10).join(f32 + f64)}
```
'''
open(os.path.join(P, "README.md"), "w").write(new)
print(f"profiles/README.md written ({len(new.splitlines())} lines)")
