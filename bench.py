"""bench.py -- throughput of the rasterizer hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one training iteration of the path BASELINE.json names: forward + backward of the tile rasterizer
for ONE view per rank (through the autograd boundary `_RasterizeGaussians`, exactly as
gaussian_renderer.render() + loss.backward() drive it), on the synthetic 500k-Gaussian / 1600x1062 scene of
SURVEY.md 8d, inputs resident in HBM.  With N > 1 ranks every rank renders a different camera of the same
replicated scene and the step ends with the RCCL exchange of parameter gradients + densification statistics
(reduced-3dgs_amd/multiview.py), i.e. weak scaling in views.

Rank 0 prints ONE JSON line: metric/value = whole-job training iterations (views) per second; plus
`roofline` for the dominant kernel (algorithmic bytes of SURVEY.md 8d / its HIP-event duration measured
inside the timed region by the library's stage timers), `cpu_baseline` (N=1 only) and `host` (what the host
side of the run looked like: CPU quota of the container, throttling during the timed region, per-step host time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "reduced-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth_scene as ss  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable
PROFILE_ROUND = next((r for r in ("r06", "r05", "r04") if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_pmc_summary.json"))), "r04")
# committed rocprofv3 --pmc summaries (tools/pmc_summary.py), per workload they were collected on
PMC_SUMMARIES = {
    "metric_500k_1600x1062": os.path.join("profiles", f"{PROFILE_ROUND}_pmc_summary.json"),
    "garden_like_2M_1600x1062": os.path.join("profiles", f"{PROFILE_ROUND}_pmc_summary_2M.json"),
    "train_like_6M_1920x1080": os.path.join("profiles", f"{PROFILE_ROUND}_pmc_summary_6M.json"),
    "clustered_500k_1600x1062": os.path.join("profiles", f"{PROFILE_ROUND}_pmc_summary_clustered_500k.json"),
}
PMC_SUMMARY = PMC_SUMMARIES["metric_500k_1600x1062"]
VALU_RATE = next((p_ for p_ in (os.path.join("profiles", n) for n in ("r06_valu_rate.txt", "r05_valu_rate.txt", "r04_valu_rate_warm.txt",
                                                                      "r03_valu_rate.txt"))
                  if os.path.exists(os.path.join(ROOT, p_))), None)
KERNEL_STATS = os.path.join("profiles", f"{PROFILE_ROUND}_kernel_stats_bench_500k_1600x1062.csv")


CLOCK_WARMUP_STEPS = 50   # untimed, ahead of the --warmup steps (see main())


def stage_bytes(P, R, N, Tn, Kbar):
    """Algorithmic bytes per launch of each stage, SURVEY.md 8d (terms of B_fwd / B_bwd regrouped by the
    library's stages; the sort term is the reference-algorithm figure R*24*ceil(bits/8) as 8d prescribes)."""
    sort_passes = 6 if Tn > 4096 else 5  # ceil((32 + msb(Tn)) / 8) for the tile counts used here
    return {
        "preprocess_fwd": P * 48 + P * 63,            # geometry kernel
        "depth_sort_scan": P * 8 + P * 12 * Kbar + P * 12,   # depth sort + scan, with the SH -> RGB stream fused in
        "tile_binning": P * 20 + R * 12 + R * 24 * sort_passes + R * 8 + Tn * 8,
        "blend_fwd": R * 40 + N * 20,
        "blend_bwd": R * 40 + N * 20 + R * 36,
        "preprocess_bwd": P * 300 + P * 92 + P * (175 + 24 * Kbar),
    }


def own_stage_bytes(P, Pb, Rb, N, Tn, Kbar, word_bytes=4, key_bytes=4, passes=2, sh_ddir=True, sparsity=False):
    """Bytes the BUILD's own algorithm has to move per launch of each stage -- the counterpart of
    stage_bytes(), which prices the REFERENCE's algorithm: no 64-bit key sort, no per-Gaussian fills, the pairs this library
    actually bins (Rb <= R, opacity-aware rects) and the Gaussians that own pairs (Pb <= P).  A stage's counter_bytes divided
    by this is its wasted traffic; alg_GBps may exceed the HBM peak, own_alg_GBps cannot.
      preprocess_fwd   read means 12 + scales 12 + rotations 16 + opacity 4; write the 48-B record, radius 4, depth key 4,
                       rect 8, tiles 4
      depth_sort_scan  histogram reads the key (4 P); scatter reads key + rect (12 P), writes one 16-B record (16 P); bucket sort
                       reads it (16 P), writes order 4 + scan 4 + depth-ordered rect 8; the colour stream reads the SH row
                       12 K + mean 12 + degree 4, writes 16 B into the record and (training forward) 36 B of direction derivatives
      tile_binning     emission reads order + rect (12 Pb) and writes pair_start (4 Pb) and the pair words; every radix pass
                       reads and writes the words, the second digit's histogram and the range kernel read the keys; ids end in
                       point_list (the last pass writes them there), 1-byte flags cleared, 8 B per tile of ranges
      blend_fwd        per list entry 4 (id) + 48 (record) once per tile + 32 B of region masks per 64 entries; per pixel 12
                       colour + 4 T + 4 n_contrib written
      blend_bwd        per list entry 4 + 48 + the masks read back; per pixel 12 dL + 4 T + 4 n_contrib read; per CONTRIBUTING
                       pair a 48-B slab row written and read by pair_reduce (counted for every binned pair: an upper bound,
                       ~45 % of them contribute) + its id 4 + flag 1; 48 B per Gaussian of sums written
      preprocess_bwd   read mean 12, record 48, sums 48, tiles 4, radius 4, scale 12, rotation 16, degree 4, direction
                       derivatives 36 (or the SH row 12 K with a sparsity term); write the nine outputs 12 K + 108"""
    return {
        "preprocess_fwd": P * 44 + P * 68,
        "depth_sort_scan": P * (4 + 12 + 16 + 16 + 16) + P * (12 * Kbar + 16 + 16 + (36 if sh_ddir else 0)),
        "tile_binning": Pb * 16 + Rb * (word_bytes + passes * 2 * word_bytes + 2 * key_bytes + 1) + Tn * 8,
        "blend_fwd": Rb * 52.5 + N * 20,
        "blend_bwd": Rb * 52.5 + N * 20 + Rb * (48 + 48 + 5) + Pb * 48,
        "preprocess_bwd": P * (148 + (12 * Kbar if sparsity else 36)) + P * (12 * Kbar + 108),
    }


# the stages bound by VALU issue: their speed of light is the calibrated issue floor of their main
# kernel (pmc_valu), not a byte count; every other stage's is the bytes it has to move at the achievable HBM rate
VALU_BOUND_STAGES = ("blend_fwd", "blend_bwd")
HBM_ACHIEVABLE_GBS = 6300.0   # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s peak


# kernels that make up each stage (name PREFIXES as rocprofv3 reports them, without arguments: the pair-word policy
# IoNarrow / IoSplit / IoWide and the digit width depend on the workload), launches per stage, and whether the kernel's
# loads are wide (16 B / lane): the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (x2) applies to those only.
STAGE_KERNELS = {
    "preprocess_fwd": [("r3::preprocess_geom_kernel<0>", 1, False)],   # <0>: the colour stream is not in this kernel
    # the SH -> RGB stream rides in spare workgroups of three of the four depth-sort kernels (preprocess.hip)
    # (the header reduction is fused into the histogram workgroups on the asynchronous path)
    "depth_sort_scan": [("r3::depth_sort_color_kernel<0, false>", 1, True),
                        ("r3::depth_colscan_kernel", 1, False), ("r3::depth_sort_color_kernel<1, false>", 1, True),
                        ("r3::depth_sort_color_kernel<2, false>", 1, True)],
    "tile_binning": [("r3::emit_pairs_kernel<", 1, False), ("r3::radix_digit_scan_kernel", 2, False),
                     ("r3::radix_scatter_kernel<", 2, False),
                     ("r3::radix_hist_kernel<", 1, False),
                     ("r3::tile_ranges_kernel<", 1, True)],
    "blend_fwd": [("r3::blend_fwd_kernel<1, false>", 1, True)],
    # (the kernel that orders the backward's (tile, segment) units heaviest first runs inside this stage's events too)
    "blend_bwd": [("r3::blend_bwd_kernel<4, true, false>", 1, True), ("r3::pair_reduce_kernel", 1, False),
                  ("r3::unit_order_kernel", 1, False)],
    # <dense degree-3 rows (M = 16: every workload here), covariance chain in double (the default; R3DGS_F64_CHAIN=0: the other)>
    "preprocess_bwd": [("r3::preprocess_bwd_kernel<true, " + ("false" if os.environ.get("R3DGS_F64_CHAIN") == "0" else "true") + ">",
                        1, False)],
}


AMBIGUOUS_KERNELS = []   # prefixes that matched several instantiations of a table (reported in the bench line)


def find_kernel(table, prefix):
    """The entry of a {kernel name: ...} table whose name is `prefix`, else the ONE entry whose name starts with it.
    Several matches (a PMC or kernel-stats file holding, say, digit widths 6 and 7 of radix_scatter_kernel, or both chain
    instantiations of preprocess_bwd_kernel) are NOT resolved by picking one -- the counters of the wrong instantiation
    would be cited silently: None is returned and the prefix is noted in AMBIGUOUS_KERNELS."""
    if prefix in table:
        return table[prefix]
    hits = [k for k in table if k.startswith(prefix)]
    if len(hits) == 1:
        return table[hits[0]]
    if len(hits) > 1:
        if prefix not in AMBIGUOUS_KERNELS:
            AMBIGUOUS_KERNELS.append(prefix)
        return None
    renamed = {"r3::unit_order_kernel": "r3::tile_order_kernel"}   # name of the same kernel in the round-4 profiles
    if prefix in renamed:
        return find_kernel(table, renamed[prefix])
    return None


def kernel_sources_sha16():
    """Fingerprint of the kernel sources (reduced-3dgs_amd/csrc/*): tools/pmc_summary.py stores it next to the counters it
    summarises, so that a bench line can say whether the counters it cites were collected on THIS build's kernels."""
    import hashlib
    d = os.path.join(ROOT, "reduced-3dgs_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_path(workload):
    p = PMC_SUMMARIES.get(workload)
    return os.path.join(ROOT, p) if p else None


def pmc_matches_build(path=None):
    """True / False: the committed PMC summary was collected on the kernel sources of this tree; None: it does not say."""
    path = path or os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    meta = json.load(open(path)).get("_meta", {})
    return meta.get("sources_sha16") == kernel_sources_sha16() if "sources_sha16" in meta else None


def pmc_traffic(stage, workload, path=None):
    """HBM bytes per launch of the stage's own kernels from the committed rocprofv3 PMC passes of this same
    command on this workload (FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, unit KiB).  gfx950 correction
    of MI355X_MICROARCH.md: FETCH_SIZE counts a 128-B read request as 64 B for wide (16 B/lane) loads, so it is doubled
    for the kernels marked wide in STAGE_KERNELS; WRITE_SIZE is taken as reported.  None if no committed counters
    match.  This is a citation of profiles/, not a measurement of the present run: see roofline.traffic_source."""
    path = path or pmc_path(workload)
    if path is None or stage not in STAGE_KERNELS or not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    total = 0.0
    for k, launches, wide in STAGE_KERNELS[stage]:
        c = find_kernel(pmc, k)
        if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None
        total += launches * ((2.0 if wide else 1.0) * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    return int(total)


# Cycles one SIMD needs per wave64 VALU instruction, measured on this chip with tools/valu_rate.hip (profiles/r03_valu_rate.txt,
# wall-clock column at 2.4 GHz, 8 waves per SIMD -- the best rate of each kind): add / mul / mov 2.4, fma 2.6 (fmac 2.8),
# v_exp / v_rcp 8.1, compares, selects, min/max, DPP, integer mad and anything with an SGPR operand 4.1-4.3.  rocprofv3
# splits SQ_INSTS_VALU into these classes (ADD_F32 also counts the DPP adds of the backward's reduction at the plain add's
# rate), so the floor below is a LOWER bound.
VALU_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.4, "SQ_INSTS_VALU_MUL_F32": 2.4, "SQ_INSTS_VALU_FMA_F32": 2.6,
               "SQ_INSTS_VALU_TRANS_F32": 8.1, "SQ_INSTS_VALU_INT32": 4.1, "SQ_INSTS_VALU_CVT": 4.1, "other": 4.1}


def committed_kernel_ms(kernel, path=None):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this command, or None."""
    import csv
    path = path or os.path.join(ROOT, KERNEL_STATS)
    if not os.path.exists(path):
        return None
    rows = {r["Name"].replace("void ", "").split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(path))}
    return find_kernel(rows, kernel)


def pmc_valu(stage, workload, stage_ms, path=None, simds=1024, ghz=2.4, kernel_ms_events=None):
    """VALU-issue view of the stage's first (main) kernel from the committed PMC passes: the two blend kernels are bound
    by VALU issue, which an HBM fraction cannot express.  floor_ms = the time 1024 SIMDs need to
    issue the kernel's VALU instructions at the per-class rates measured by tools/valu_rate.hip.  The stage timer of this
    run covers the stage's other kernels too (pair_reduce and the tile order for blend_bwd), so kernel_ms = this run's
    stage time minus those kernels' committed averages (the committed average of the kernel itself is given beside it)
    and frac = floor / kernel_ms; frac_of_stage = floor / the stage time as measured."""
    path = path or os.path.join(ROOT, PMC_SUMMARY)
    if workload != "metric_500k_1600x1062" or stage not in STAGE_KERNELS or not os.path.exists(path):
        return None
    k = STAGE_KERNELS[stage][0][0]
    c = find_kernel(json.load(open(path)), k) or {}
    if "SQ_INSTS_VALU" not in c:
        return None
    classes = {n: c.get(n, 0.0) for n in VALU_CYCLES if n != "other"}
    other = max(0.0, c["SQ_INSTS_VALU"] - sum(classes.values()))
    cycles = sum(VALU_CYCLES[n] * v for n, v in classes.items()) + VALU_CYCLES["other"] * other
    floor_ms = cycles / simds / (ghz * 1e9) * 1e3
    others = [committed_kernel_ms(n) for n, _, _ in STAGE_KERNELS[stage][1:]]
    committed = committed_kernel_ms(k)
    kernel_ms = stage_ms - sum(others) if (stage_ms and all(o is not None for o in others)) else stage_ms
    if kernel_ms_events:   # round 6: the kernel alone between its own HIP events of THIS run (stage blend_bwd_kernel)
        kernel_ms = kernel_ms_events
    out = {"kernel": k, "insts": int(c["SQ_INSTS_VALU"]),
           "insts_by_class": {n.replace("SQ_INSTS_VALU_", "").lower(): int(v) for n, v in classes.items()} | {"other": int(other)},
           "cycles_per_class": {n.replace("SQ_INSTS_VALU_", "").lower(): v for n, v in VALU_CYCLES.items()},
           "floor_ms": round(floor_ms, 4), "stage_ms": round(stage_ms, 4), "kernel_ms": round(kernel_ms, 4),
           "kernel_ms_committed_profile": round(committed, 4) if committed else None,
           "frac": round(floor_ms / kernel_ms, 3) if kernel_ms else None,
           "frac_of_stage": round(floor_ms / stage_ms, 3) if stage_ms else None,
           "kernel_ms_source": "HIP events around the kernel alone, this run" if kernel_ms_events else
                               "this run's stage time minus the other kernels' committed averages",
           "assumes": f"{simds} SIMDs at {ghz} GHz", "source": f"{PMC_SUMMARY}, {VALU_RATE}"}
    if "SQ_ACTIVE_INST_VALU" in c:   # what the SQ itself reports per instruction (quad-cycles, per wave: >= 4)
        out["sq_active_cycles_per_inst"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"], 2)
    return out


def cgroup_cpu():
    """CPU bandwidth limit and throttling counters of this container (cgroup v2), or None.  The GPU boxes give a job
    256 visible CPUs but a quota of a few: a host side that spins or fans out threads gets frozen for the rest of
    each 100 ms period, and a 20-step timed region is shorter than one freeze."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        stat = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return {"quota_cpus": None if quota == "max" else round(int(quota) / int(period), 2),
                "nr_throttled": int(stat.get("nr_throttled", 0)), "throttled_usec": int(stat.get("throttled_usec", 0)),
                "usage_usec": int(stat.get("usage_usec", 0))}
    except Exception:
        return None


def wait_for_cfs_period_start(max_wait_s=0.25):
    """Blocks until the container's CPU-bandwidth period has just rolled over (cpu.stat nr_periods changes), so that
    a short timed region starts with a full CPU quota instead of running into the freeze at the end of a period that
    other processes of the container have already spent.  Returns True if a rollover was seen."""
    def periods():
        try:
            for line in open("/sys/fs/cgroup/cpu.stat"):
                if line.startswith("nr_periods"):
                    return int(line.split()[1])
        except Exception:
            pass
        return None
    p0 = periods()
    if p0 is None:
        return False
    t_end = time.perf_counter() + max_wait_s
    while time.perf_counter() < t_end:
        time.sleep(0.0005)
        if periods() != p0:
            return True
    return False


def effective_cpus():
    cg = cgroup_cpu()
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if cg and cg["quota_cpus"]:
        n = max(1, min(n, int(cg["quota_cpus"])))
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="metric_500k_1600x1062", choices=list(ss.WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cameras", type=int, default=8)
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed region and its instrumented pass: no render-only, lambda_sh_sparsity or reference-mode "
                         "leg -- for rocprofv3 passes, whose per-kernel means would otherwise mix those legs' launches in")
    args = ap.parse_args()
    cg_warm0 = cgroup_cpu()   # throttling seen from here to the timed region decides whether to align with a period
    ncpu = effective_cpus()
    torch.set_num_threads(ncpu)   # the container's CPU quota, not the 256 visible CPUs (nothing timed is CPU-parallel)
    # run autograd's backward in the calling thread: no hand-off to a per-device worker thread per iteration
    torch.autograd.set_multithreading_enabled(False)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # R3DGS_BENCH_SINGLE_DEVICE=1 (+ R3DGS_BENCH_BACKEND=gloo): functional dry run of the N>1 code path on a box with
    # one GPU (all ranks share cuda:0, exchange through gloo) -- never a performance number
    same_dev = os.environ.get("R3DGS_BENCH_SINGLE_DEVICE") == "1"
    dev_index = 0 if (world == 1 or same_dev) else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        backend = os.environ.get("R3DGS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = torch.device("cuda", dev_index)

    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    from multiview import ViewParallelExchange

    w = ss.WORKLOADS[args.workload]
    W, H, P = w["W"], w["H"], w["P"]
    N, Tn = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    _, cam0, g = ss.make_workload(args.workload, seed=0)
    Kbar = float(((g["degrees"].reshape(-1) + 1) ** 2).mean())

    def dv(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    leaves = {k: dv(g[k]).requires_grad_() for k in ("means3D", "opacity", "scales", "rotations", "sh")}
    degrees = dv(g["degrees"])
    bg = dv(np.zeros(3, np.float32))
    dl = dv(ss.upstream_grad(W, H, seed=1))
    empty = torch.Tensor([])
    cams = [cam0] + [ss.make_camera(W, H, w["f"], s) for s in range(1, args.cameras)]
    settings = [dgr.GaussianRasterizationSettings(H, W, c.tanfovx, c.tanfovy, bg, 1.0, dv(c.world_view_transform),
                                                  dv(c.full_proj_transform), 3, dv(c.camera_center), False, False)
                for c in cams]
    # all-to-all + local combine + all-gather needs RCCL; the single-device gloo dry run falls back to all-reduces
    two_phase = os.environ.get("R3DGS_BENCH_BACKEND", "nccl") == "nccl"
    GRAD_SHAPES = {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}
    exch = ViewParallelExchange(GRAD_SHAPES, P, device, two_phase=two_phase) if world > 1 else None
    if exch is not None and os.environ.get("R3DGS_BENCH_NO_ARENA") != "1":
        _C.set_gradient_arena(exch.arena)   # parameter gradients are written straight into the exchange buffer

    born_in_buffer = [None]
    pending = [None]     # exchange of the previous step still in flight
    # N > 1: the timed region (-> `value`) runs the SERIALISED exchange: step k's sums are complete before step k+1
    # starts, i.e. "all-reduce before the optimizer / prune step" as BASELINE.json words it.  The overlapped form (step
    # k's buffer travels while step k+1 renders; its sums would be applied one step late) is measured afterwards and
    # reported as `value_overlapped`.
    overlap = [False]

    def cam_index(step):
        return (step * world + rank) % len(cams)

    def train_step(step, lam=0.0):
        for t in leaves.values():
            t.grad = None
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0  # gaussian_renderer/__init__.py:27
        means2D.retain_grad()
        color, radii = dgr.rasterize_gaussians(leaves["means3D"], means2D, leaves["sh"], degrees, empty,
                                               leaves["opacity"], leaves["scales"], leaves["rotations"], empty,
                                               settings[cam_index(step)], lam)
        color.backward(dl)
        if exch is not None:
            grads = {k: v.grad for k, v in leaves.items()}
            if step == 0:   # reported once: did autograd keep the arena views as .grad (zero-copy pack)?
                born_in_buffer[0] = sum(exch.arena(k, tuple(g_.shape)) is not None and
                                        g_.data_ptr() == exch.arena(k, tuple(g_.shape)).data_ptr()
                                        for k, g_ in grads.items())
            exch.pack(grads, means2D.grad, radii)
            if overlap[0]:
                # The exchange of step k travels while step k+1 renders (double-buffered arena, own stream); the
                # optimizer of step k would run where the wait is, one step later.
                if pending[0] is not None:
                    exch.wait(pending[0])
                pending[0] = exch.exchange_async()
            else:
                exch.exchange()
        return radii

    def drain():
        if exch is not None and pending[0] is not None:
            exch.wait(pending[0])
            pending[0] = None

    # num_rendered per camera (property of the input; every per-pair byte term scales with it); these passes also
    # teach the library the pair reservation of this view size
    Rs, Vs, Ps = [], [], []
    with torch.no_grad():
        for s_ in settings:
            out = _C.rasterize_gaussians(s_.bg, leaves["means3D"], empty, leaves["opacity"], leaves["scales"],
                                         leaves["rotations"], 1.0, empty, s_.viewmatrix, s_.projmatrix, s_.tanfovx,
                                         s_.tanfovy, H, W, leaves["sh"], degrees, s_.campos, False, False)
            Rs.append(int(out[0]))          # the reference's num_rendered: what the byte formulas of SURVEY 8d are written in
            Ps.append(out[0].pairs)         # pairs this library bins (opacity-aware rects leave unreachable tiles out)
            Vs.append(int((out[2] > 0).sum()))

    def barrier():
        if world > 1:
            dist.barrier()

    # The GPU's clocks take a few tens of milliseconds of load to settle (20 timed steps after 5 warm-up steps measured
    # 1136-1160 it/s, after 50: 1162-1164): CLOCK_WARMUP_STEPS untimed steps of the same work run ahead of the W
    # warm-up steps, whatever W is.  Reported as config.clock_warmup_steps; the timed region is exactly K steps.
    for i in range(CLOCK_WARMUP_STEPS):
        train_step(i)
    for i in range(args.warmup):
        train_step(i)
    drain()
    torch.cuda.synchronize()
    # Timed region: HIP events around the DOMINANT stage only.  A pass with a timed stage is issued with direct
    # launches (the events sit between its kernels) instead of its graph, so the full per-stage breakdown is taken in
    # a second, untimed instrumented pass below and only the backward blend is timed here.
    dom_stage = os.environ.get("R3DGS_BENCH_DOM_STAGE", "blend_bwd")
    prof_mode = os.environ.get("R3DGS_BENCH_PROFILE", "dominant")  # dominant | all | off  (A/B of the timer cost)
    # blend_bwd_kernel: the backward blend kernel ALONE (nested inside the blend_bwd stage's events) -> roofline.kernel_frac
    # (the kernel-only timer `blend_bwd_kernel` -> roofline.kernel_frac runs in the instrumented pass BEHIND the timed region:
    # two more event packets between the backward's kernels cost the timed step ~5 us, measured in round 6's first visit)
    timed_stages = [dom_stage]
    if prof_mode == "all":
        _C.profile_enable(True)
    elif prof_mode == "dominant":
        _C.profile_enable(True, only=timed_stages)
    _C.profile_read()
    overflow0 = _C.reserve_overflow_events()
    stats0 = _C.pass_stats()
    barrier()
    torch.cuda.synchronize()
    # If this container is being CPU-throttled right now (someone in it -- not this process, which uses ~1 CPU -- burns
    # the quota: the cgroup then freezes for the rest of every 100 ms period), start the timed region right after a
    # period rollover.  The freeze would otherwise land inside a 20-step region and be reported as step time.
    cg_w = cgroup_cpu()
    cfs_aligned = False
    if cg_w and cg_warm0 and cg_w["nr_throttled"] > cg_warm0["nr_throttled"] and world == 1:
        cfs_aligned = wait_for_cfs_period_start()
    cg0 = cgroup_cpu()
    host_ms = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        th = time.perf_counter()
        train_step(args.warmup + i)
        host_ms.append(1e3 * (time.perf_counter() - th))
    drain()
    ev1.record()
    t_enq = time.perf_counter()
    # The host is done long before the GPU.  Wait without burning the container's CPU quota (a spinning
    # synchronize() is one fully busy CPU): poll the end event between short sleeps, then synchronize.
    while not ev1.query():
        time.sleep(5e-5)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    cg1 = cgroup_cpu()
    gpu_event_ms = ev0.elapsed_time(ev1)
    stats1 = _C.pass_stats()
    overflow1 = _C.reserve_overflow_events()
    prof_timed = _C.profile_read()
    _C.profile_enable(False)
    # second, untimed pass with every stage timer on: the per-stage breakdown
    _C.profile_enable(True)
    for i in range(min(args.steps, 20)):
        train_step(args.warmup + i)
    drain()
    torch.cuda.synchronize()
    prof = _C.profile_read()
    _C.profile_enable(False)
    # N > 1: the exchange alone (events around a synchronous exchange), whether every replica ended up with the same
    # bits, and the same K steps in the overlapped (one-step-late) form
    exchange_ms = overlapped_ms_per_step = replicas_identical = exchange_forms = None
    if exch is not None:
        n_x = min(args.steps, 10)
        xa, xb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        xa.record()
        for _ in range(n_x):
            exch.exchange()
        xb.record()
        torch.cuda.synchronize()
        exchange_ms = xa.elapsed_time(xb) / n_x
        # one more serialised step, then compare a checksum of the exchanged buffer (sums, statistics, radii) across ranks:
        # the combine runs in rank order on every rank, so the replicas must agree bit for bit
        train_step(args.warmup)
        torch.cuda.synchronize()
        bits = exch.flat[:exch.total].view(torch.int32).to(torch.int64)
        mine = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=device) % 8191 + 1)).sum()])
        if os.environ.get("R3DGS_BENCH_BACKEND", "nccl") != "nccl":
            mine = mine.cpu()   # gloo gathers host tensors
        allsums = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allsums, mine)
        replicas_identical = all(bool(torch.equal(allsums[0], t)) for t in allsums)
        assert replicas_identical, "view-parallel exchange: the ranks' summed gradients differ"
        # The compact transports (multiview.py, round 6): visible-union rows and / or bfloat16 for the higher SH bands --
        # each measured as the dense form was (events around synchronous exchanges of a freshly packed step) with its bytes
        # and the model's prediction beside it.  `value` stays the dense fp32 form: north_star's all-reduce.  Wrapped: a
        # failure here must not cost the run its headline number.
        dense_ex = exch
        try:
            from multiview import exchange_model
            step_compute_ms = max(1e3 * elapsed / args.steps - exchange_ms, 1e-3)
            exchange_forms = {"dense_fp32": {"exchange_ms": round(exchange_ms, 4), "bytes_per_rank": exch.bytes_per_rank(),
                                             "rows": P, "model": exchange_model(P, world, step_compute_ms)}}
            rest_floats = 3.0 * (Kbar - 1.0)   # mean higher-band SH floats a row carries when the bands travel by degree
            for tag, kw, by_degree in (("visible_union", dict(sparse=True), False), ("bf16_sh_rest", dict(sh_rest_bf16=True), False),
                                       ("visible_union_bf16_sh_rest", dict(sparse=True, sh_rest_bf16=True), False),
                                       ("sh_bands_by_degree", dict(), True),
                                       ("sh_bands_by_degree_bf16", dict(sh_rest_bf16=True), True)):
                ex2 = ViewParallelExchange(GRAD_SHAPES, P, device, two_phase=True, **kw)
                if by_degree:
                    ex2.set_degrees(degrees)
                exch = ex2
                if os.environ.get("R3DGS_BENCH_NO_ARENA") != "1":
                    _C.set_gradient_arena(ex2.arena)
                times = []
                for i in range(min(args.steps, 6)):
                    for t_ in leaves.values():
                        t_.grad = None
                    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0
                    m2.retain_grad()
                    c_, r_ = dgr.rasterize_gaussians(leaves["means3D"], m2, leaves["sh"], degrees, empty, leaves["opacity"],
                                                     leaves["scales"], leaves["rotations"], empty,
                                                     settings[cam_index(args.warmup + i)], 0.0)
                    c_.backward(dl)
                    ex2.pack({k: v.grad for k, v in leaves.items()}, m2.grad, r_)
                    barrier()
                    torch.cuda.synchronize()
                    xa.record()
                    ex2.exchange()
                    xb.record()
                    torch.cuda.synchronize()
                    if i:
                        times.append(xa.elapsed_time(xb))
                t_ms = torch.tensor([sum(times) / max(len(times), 1)], dtype=torch.float64, device=device)
                dist.all_reduce(t_ms, op=dist.ReduceOp.MAX)
                rows = int(ex2.last["rows"])
                exchange_forms[tag] = {"exchange_ms": round(float(t_ms.item()), 4), "bytes_per_rank": ex2.last["bytes"],
                                       "rows": rows, "form": ex2.last["form"],
                                       "model": exchange_model(P, world, step_compute_ms, union_frac=rows / P,
                                                               sparse=bool(kw.get("sparse")),
                                                               sh_rest_bf16=bool(kw.get("sh_rest_bf16")),
                                                               sh_rest_floats=rest_floats if by_degree else 45.0)}
                exchange_forms[tag]["serialised_iters_per_s_with_this_form"] = round(
                    world / ((step_compute_ms + exchange_forms[tag]["exchange_ms"]) * 1e-3), 2)
            exch = dense_ex
            if os.environ.get("R3DGS_BENCH_NO_ARENA") != "1":
                _C.set_gradient_arena(exch.arena)
        except Exception as e:   # noqa: BLE001
            exchange_forms = {"error": repr(e)}
            exch = dense_ex
            if os.environ.get("R3DGS_BENCH_NO_ARENA") != "1":
                _C.set_gradient_arena(exch.arena)
        overlap[0] = True
        for i in range(3):
            train_step(i)
        drain()
        barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for i in range(args.steps):
            train_step(args.warmup + i)
        drain()
        torch.cuda.synchronize()
        barrier()
        t_o = torch.tensor([time.perf_counter() - ts0], dtype=torch.float64, device=device)
        dist.all_reduce(t_o, op=dist.ReduceOp.MAX)
        overlapped_ms_per_step = 1e3 * float(t_o.item()) / args.steps
        overlap[0] = False
    for st_ in timed_stages:
        if prof_timed.get(st_, (0, 0))[1]:
            prof[st_] = prof_timed[st_]  # the roofline kernel's time is the one from the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # render-only throughput (render.py's FPS path: forward under no_grad), after the timed region
    with torch.no_grad():
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for i in range(0 if args.main_only else args.steps):
            s_ = settings[cam_index(i)]
            _C.rasterize_gaussians(s_.bg, leaves["means3D"], empty, leaves["opacity"], leaves["scales"],
                                   leaves["rotations"], 1.0, empty, s_.viewmatrix, s_.projmatrix, s_.tanfovx,
                                   s_.tanfovy, H, W, leaves["sh"], degrees, s_.campos, False, False)
        torch.cuda.synchronize()
        render_s = time.perf_counter() - tr0

    # The method's own configuration (full_eval.py:33,44: every paper run trains with --lambda_sh_sparsity=0.1): the same K
    # steps with the SH L1 term on -- the per-Gaussian backward then reads the SH rows (signs of the coefficients) instead
    # of the direction derivatives the forward left.  A second number; `value` stays lambda = 0,
    # the default of arguments/__init__.py:94.
    sparsity = None
    if world == 1 and not args.main_only:
        lam = 0.1
        for i in range(5):
            train_step(i, lam)
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for i in range(args.steps):
            train_step(args.warmup + i, lam)
        e1 = torch.cuda.Event()
        e1.record()
        while not e1.query():
            time.sleep(5e-5)
        torch.cuda.synchronize()
        el_s = time.perf_counter() - ts0
        _C.profile_enable(True)
        _C.profile_read()
        for i in range(min(args.steps, 20)):
            train_step(args.warmup + i, lam)
        torch.cuda.synchronize()
        prof_s = _C.profile_read()
        _C.profile_enable(False)
        sparsity = {"lambda_sh_sparsity": lam, "value": round(args.steps / el_s, 2), "unit": "iters/s",
                    "ms_per_step": round(1e3 * el_s / args.steps, 4),
                    "stages_ms": {k: round(ms / cnt, 4) for k, (ms, cnt) in prof_s.items() if cnt}}

    # The configuration whose RESULTS are the reference's: the reference's 3-sigma tile squares
    # (set_tight_rects(False): lists identical to rasterizer_impl.cu:78-117's) and its fp32 covariance chain
    # (set_f64_chain(False): backward.cu:228-306, 311-374).  The same K steps, the same cameras; a second number, `value`
    # stays the default mode.  (the driver's record then holds the rate of the mode a CUDA build can be
    # compared with at 1e-4.)
    reference_mode = None
    if world == 1 and not args.main_only:
        was_tight, was_f64 = _C.set_tight_rects(False), _C.set_f64_chain(False)
        try:
            with torch.no_grad():   # the longer lists' pair counts, per camera (and the reservation learns them)
                for s_ in settings:
                    _C.rasterize_gaussians(s_.bg, leaves["means3D"], empty, leaves["opacity"], leaves["scales"],
                                           leaves["rotations"], 1.0, empty, s_.viewmatrix, s_.projmatrix, s_.tanfovx,
                                           s_.tanfovy, H, W, leaves["sh"], degrees, s_.campos, False, False)
            for i in range(max(args.warmup, 2 * len(settings))):
                train_step(i)
            torch.cuda.synchronize()
            stats_r0 = _C.pass_stats()
            ts0 = time.perf_counter()
            for i in range(args.steps):
                train_step(args.warmup + i)
            e1 = torch.cuda.Event()
            e1.record()
            while not e1.query():
                time.sleep(5e-5)
            torch.cuda.synchronize()
            el_r = time.perf_counter() - ts0
            stats_r1 = _C.pass_stats()
            _C.profile_enable(True)
            _C.profile_read()
            for i in range(min(args.steps, 20)):
                train_step(args.warmup + i)
            torch.cuda.synchronize()
            prof_r = _C.profile_read()
            _C.profile_enable(False)
            reference_mode = {"tight_rects": _C.tight_rects(), "f64_covariance_chain": _C.f64_chain(),
                              "value": round(args.steps / el_r, 2), "unit": "iters/s",
                              "ms_per_step": round(1e3 * el_r / args.steps, 4),
                              "passes": {k: stats_r1[k] - stats_r0[k] for k in ("reserved_passes", "exact_passes", "redone_passes")},
                              "stages_ms": {k: round(ms / cnt, 4) for k, (ms, cnt) in prof_r.items() if cnt},
                              "what": "same K steps and cameras with set_tight_rects(False) + set_f64_chain(False): the "
                                      "reference's tile lists and fp32 covariance chain (the mode to compare a CUDA build with)"}
        finally:
            _C.set_tight_rects(was_tight)
            _C.set_f64_chain(was_f64)

    used = [cam_index(args.warmup + i) for i in range(args.steps)]
    R_mean = float(np.mean([Rs[k] for k in used]))
    V_mean = float(np.mean([Vs[k] for k in used]))
    pairs_mean = float(np.mean([Ps[k] for k in used]))
    sb = stage_bytes(P, R_mean, N, Tn, Kbar)
    id_bits, tile_bits = max(1, int(np.ceil(np.log2(max(P, 2))))), max(1, int(np.ceil(np.log2(max(Tn, 2)))))
    word_bytes, key_bytes = (4, 4) if id_bits + tile_bits <= 32 else ((6, 2) if Tn <= 65536 else (8, 8))
    own = own_stage_bytes(P, V_mean, pairs_mean, N, Tn, Kbar, word_bytes, key_bytes,
                          passes=2 if tile_bits <= 14 else -(-tile_bits // 8))
    kernel_ms_events = None   # the backward blend kernel alone, HIP events of the instrumented pass
    if prof.get("blend_bwd_kernel", (0, 0))[1]:
        kernel_ms_events = prof["blend_bwd_kernel"][0] / prof["blend_bwd_kernel"][1]
    stages = {}
    for name, (ms, cnt) in prof.items():
        if cnt and name != "blend_bwd_kernel":
            avg_ms = ms / cnt
            b = sb.get(name, 0)   # "sh_color" only exists as a stage of its own on the generic-sort path
            # alg_GBps: SURVEY 8d's reference-algorithm bytes of the stage / its time.  NOT a bandwidth: bytes the library
            # never moves (the reference's 64-bit key sort, its per-Gaussian fills) count, so it can exceed the HBM peak;
            # counter_GBps: the bytes the committed rocprofv3 counters of this workload saw move / this run's time
            tr = pmc_traffic(name, args.workload)
            ob = own.get(name)
            stages[name] = {"avg_ms": round(avg_ms, 4), "launches": cnt, "alg_bytes": int(b),
                            "alg_GBps": round(b / (avg_ms * 1e-3) / 1e9, 1),
                            # own_alg_*: the bytes THIS build's algorithm has to move (own_stage_bytes: pairs binned, no 64-bit
                            # sort, no fills) -- a rate that cannot exceed the HBM peak; alg_* is the reference's algorithm
                            "own_alg_bytes": int(ob) if ob is not None else None,
                            "own_alg_GBps": round(ob / (avg_ms * 1e-3) / 1e9, 1) if ob is not None else None,
                            "counter_bytes": tr,
                            "counter_GBps": round(tr / (avg_ms * 1e-3) / 1e9, 1) if tr is not None else None,
                            "counter_over_own_alg": round(tr / ob, 2) if (tr is not None and ob) else None}
    dom = max(stages, key=lambda k: stages[k]["avg_ms"]) if stages else None
    roofline = None
    if dom:
        A = stages[dom]["alg_GBps"]
        traffic = pmc_traffic(dom, args.workload)
        # kernel_frac: SURVEY 8d's bytes of the dominant KERNEL alone (R * 76 + N * 20 for the backward blend) over the
        # kernel's own duration between HIP events of this run's timed region -- what a reader recomputes from the
        # rocprofv3 kernel trace; frac: the whole stage's bytes over the stage's events (unit order + kernel + pair reduction)
        kernel_alg = R_mean * 76 + N * 20 if dom == "blend_bwd" else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": A, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(A / HBM_PEAK_GBS, 4),
                    "kernel_name": STAGE_KERNELS[dom][0][0] if dom in STAGE_KERNELS else None,
                    "kernel_ms": round(kernel_ms_events, 4) if (kernel_ms_events and dom == "blend_bwd") else None,
                    "kernel_alg_bytes": int(kernel_alg) if kernel_alg else None,
                    "kernel_achieved": (round(kernel_alg / (kernel_ms_events * 1e-3) / 1e9, 1)
                                        if (kernel_alg and kernel_ms_events and dom == "blend_bwd") else None),
                    "kernel_frac": (round(kernel_alg / (kernel_ms_events * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                    if (kernel_alg and kernel_ms_events and dom == "blend_bwd") else None),
                    "traffic": traffic,
                    "traffic_source": (PMC_SUMMARIES[args.workload] + " (committed rocprofv3 --pmc passes of this command; "
                                       "not collected in this run)") if traffic is not None else None,
                    "traffic_collected_on_this_build": pmc_matches_build(pmc_path(args.workload)) if traffic is not None else None,
                    "duration_source": "HIP events around the stage on its stream, inside the timed region",
                    "kernel_ms_source": "HIP events around the kernel alone, instrumented pass of this run (behind the timed region)",
                    "valu": pmc_valu(dom, args.workload, stages[dom]["avg_ms"],
                                     kernel_ms_events=kernel_ms_events if dom == "blend_bwd" else None)}
    iters_per_s = args.steps * world / elapsed
    B_iter = P * (718 + 36 * Kbar) + R_mean * 280 + N * 40
    gpu_ms = sum(v["avg_ms"] for v in stages.values())
    per_stage_traffic = [pmc_traffic(k, args.workload) for k in STAGE_KERNELS]
    counter_bytes = int(sum(per_stage_traffic)) if all(t is not None for t in per_stage_traffic) else None
    # speed of light of this build's own algorithm: the VALU-bound stages at the calibrated issue floor of their main
    # kernel, every other stage at the bytes it has to move (own_stage_bytes) over the achievable HBM rate
    sol_parts, sol_ok = {}, bool(stages)
    for name in stages:
        if name in VALU_BOUND_STAGES:
            v_ = pmc_valu(name, args.workload, stages[name]["avg_ms"])
            if v_ is None:
                sol_ok = False
            else:
                sol_parts[name] = v_["floor_ms"]
        elif stages[name]["own_alg_bytes"] is not None:
            sol_parts[name] = stages[name]["own_alg_bytes"] / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
    sol_ms = sum(sol_parts.values()) if sol_ok else None
    hs = sorted(host_ms)
    host = {"cpus_visible": os.cpu_count(), "cpus_effective": ncpu,
            "loadavg": [round(x, 1) for x in os.getloadavg()],
            "host_ms_per_step_min_med_max": [round(hs[0], 3), round(hs[len(hs) // 2], 3), round(hs[-1], 3)],
            "enqueue_ms_total": round(1e3 * (t_enq - t0), 3),
            "gpu_event_ms_per_step": round(gpu_event_ms / args.steps, 4),   # first to last kernel of the timed region
            "gpu_stage_ms_sum": round(gpu_ms, 4),
            "step_over_gpu_stage_sum": round(1e3 * elapsed / args.steps / gpu_ms, 3) if gpu_ms else None}
    if cg0 and cg1:
        host["cgroup"] = {"quota_cpus": cg1["quota_cpus"],
                          "throttled_periods_in_timed_region": cg1["nr_throttled"] - cg0["nr_throttled"],
                          "throttled_ms_in_timed_region": round((cg1["throttled_usec"] - cg0["throttled_usec"]) / 1e3, 2),
                          "cpu_ms_in_timed_region": round((cg1["usage_usec"] - cg0["usage_usec"]) / 1e3, 2),
                          "throttled_periods_before_timed_region": cg_w["nr_throttled"] - cg_warm0["nr_throttled"] if cg_w and cg_warm0 else None,
                          "timed_region_started_at_period_rollover": cfs_aligned}
    result = {
        "metric": "train iters/s (fwd+bwd) + Mpix/s render, 500k Gaussians @1600x1062",
        "value": round(iters_per_s, 2), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "gaussians": P, "width": W, "height": H,
                   "sh_degree": {"all3": 3, "all0": 0}.get(w["degree_mode"], "mixed 0-3"), "sh_coeffs_mean": round(Kbar, 2),
                   "views_per_step": world, "visible_mean": round(V_mean), "num_rendered_mean": round(R_mean),
                   "pairs_binned_mean": round(pairs_mean), "tight_rects": _C.tight_rects(),
                   "binding": _C.binding(), "f64_covariance_chain": _C.f64_chain(),
                   "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU",
                   "exchange": ("all-to-all + local SUM/MAX combine + all-gather of one flat buffer (59 fp32 grads + 2 "
                                "stats + radii per Gaussian) after every step's backward, complete before the next step "
                                "starts (value / value_serialised); value_overlapped: the same exchange on its own "
                                "stream, double-buffered, travelling while the next step renders (sums one step late)")
                   if world > 1 else None,
                   "replicas_bit_identical": replicas_identical,
                   "grads_born_in_exchange_buffer": born_in_buffer[0],
                   "clock_warmup_steps": CLOCK_WARMUP_STEPS,
                   "issue": "one hipGraph launch per forward, direct launches for the (event-timed) backward; "
                            "pair reservation (per camera) instead of a num_rendered read-back; strict: every forward "
                            "checks its published pair count before returning and redoes an overflowed pass exactly",
                   "strict": _C.is_strict(),
                   "reserve_overflows_in_run": overflow1 - overflow0,
                   "passes_in_timed_region": {k: stats1[k] - stats0[k] for k in
                                              ("reserved_passes", "exact_passes", "redone_passes")}},
        "value_serialised": round(iters_per_s, 2) if world > 1 else None,
        "value_overlapped": round(args.steps * world / (overlapped_ms_per_step * 1e-3 * args.steps), 2)
        if overlapped_ms_per_step else None,
        "exchange_ms": round(exchange_ms, 4) if exchange_ms is not None else None,
        # bytes ONE rank sends per step (both phases, all peers); exchange_forms: the same for the opt-in compact transports,
        # measured the same way, each with the bytes / time model's prediction (multiview.exchange_model)
        "exchange_bytes_per_rank": exch.bytes_per_rank() if exch is not None else None,
        "exchange_forms": exchange_forms,
        "step_ms_exchange_overlapped": round(overlapped_ms_per_step, 4) if overlapped_ms_per_step is not None else None,
        "overlap_frac": (round(max(0.0, min(1.0, (1e3 * elapsed / args.steps - overlapped_ms_per_step) / exchange_ms)), 3)
                         if exchange_ms and overlapped_ms_per_step else None),
        "render_mpix_per_s": round(args.steps * N / render_s / 1e6, 1) if not args.main_only else None,
        "render_fps": round(args.steps / render_s, 1) if not args.main_only else None,
        "roofline": roofline,
        # frac_of_8TBps: SURVEY 8d's reference-algorithm bytes (work this library avoids counts in its favour) -- NOT HBM
        # utilisation; frac_counter_traffic: the bytes the committed rocprofv3 counters saw move, per step, over the same time
        "iter_roofline": {"B_iter_bytes": int(B_iter), "achieved_GBps": round(B_iter * iters_per_s / world / 1e9, 1),
                          "frac_of_8TBps": round(B_iter * iters_per_s / world / 8e12, 4),
                          "counter_traffic_bytes": counter_bytes,
                          "frac_counter_traffic": (round(counter_bytes * iters_per_s / world / 8e12, 4)
                                                   if counter_bytes else None),
                          "counter_traffic_source": PMC_SUMMARIES.get(args.workload) if counter_bytes else None,
                          # sum over the stages of: the VALU issue floor (blend kernels) / own-algorithm bytes at 6.3 TB/s
                          "speed_of_light_ms": round(sol_ms, 4) if sol_ms else None,
                          "speed_of_light_parts_ms": {k: round(v, 4) for k, v in sol_parts.items()} if sol_ms else None,
                          "frac_of_speed_of_light": round(sol_ms / (1e3 * elapsed / args.steps), 3) if sol_ms else None,
                          "own_alg_bytes": int(sum(v["own_alg_bytes"] or 0 for v in stages.values())),
                          "speed_of_light_assumes": f"{HBM_ACHIEVABLE_GBS / 1e3} TB/s for the HBM-bound stages; "
                                                    f"VALU floors from {PMC_SUMMARIES.get(args.workload)}"},
        "value_sh_sparsity": sparsity["value"] if sparsity else None,
        "sh_sparsity": sparsity,
        "value_reference_mode": reference_mode["value"] if reference_mode else None,
        "reference_mode": reference_mode,
        "ambiguous_profile_kernels": AMBIGUOUS_KERNELS or None,
        "stages": stages,
        "stages_note": f"{dom_stage}: HIP events inside the timed region; other stages: separate instrumented pass",
        "host": host,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload, W, H, g, cams[0], ncpu)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(workload, W, H, g, cam, ncpu):
    """the "PyTorch CPU autograd reference render" of BASELINE.json configs[0] (10k Gaussians, 400x400,
    degree 0) -- the repo's fp32 PyTorch restatement (oracle/torch_ref.py, tile by tile) with
    torch.set_num_threads(n) on this box's host cores, forward + backward, median of 5 after one warm-up -- and,
    as a second sample, ONE iteration of the bench workload itself by the C restatement (OpenMP in the per-pixel /
    per-tile stages).  Both limited to the container's CPU quota.  A reported baseline, not a target."""
    import statistics
    os.environ["OMP_NUM_THREADS"] = str(ncpu)
    from oracle import oracle as orc
    from oracle import torch_ref as tr
    w0 = ss.WORKLOADS["cfg0_10k_400"]
    cam0 = ss.make_camera(w0["W"], w0["H"], w0["f"], None)
    g0 = ss.make_gaussians(w0["P"], cam0, seed=0, degree_mode=w0["degree_mode"])
    dl0 = torch.tensor(ss.upstream_grad(w0["W"], w0["H"], seed=1))
    f32 = torch.float32

    def tt(a):
        return torch.tensor(np.asarray(a), dtype=f32)

    times = []
    for rep in range(6):
        leaves = [tt(g0[k]).requires_grad_() for k in ("means3D", "opacity", "scales", "rotations", "sh")]
        t0 = time.perf_counter()
        color, _, _ = tr.render(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], torch.tensor(g0["degrees"]),
                                tt(cam0.world_view_transform), tt(cam0.full_proj_transform), tt(cam0.camera_center),
                                torch.zeros(3), w0["W"], w0["H"], cam0.tanfovx, cam0.tanfovy, tiled=True)
        (color * dl0).sum().backward()
        if rep:
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    dl_np = ss.upstream_grad(W, H, seed=1)
    tc0 = time.perf_counter()
    ref = orc.forward(np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0,
                      None, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, g["sh"],
                      g["degrees"], cam.camera_center)
    tc1 = time.perf_counter()
    orc.backward(ref["state"], dl_np, 0.0)
    tc2 = time.perf_counter()
    cpu_model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": round(1.0 / med, 4), "unit": "iters/s", "cores": ncpu, "cpu_model": cpu_model, "kind": "port",
            "threads_effective": torch.get_num_threads(),
            "sample": f"configs[0] (10k Gaussians, 400x400, degree 0): fwd+bwd of the fp32 PyTorch restatement "
                      f"(oracle/torch_ref.py, tiled), torch.set_num_threads({torch.get_num_threads()}), median of 5 "
                      f"after 1 warm-up = {med:.3f} s [{min(times):.3f}, {max(times):.3f}]",
            "same_workload_sample": {"value": round(1.0 / (tc2 - tc0), 4), "unit": "iters/s",
                                     "what": f"1 iteration of {workload} by the C restatement (oracle/raster_oracle.c): fwd "
                                             f"{tc1 - tc0:.2f} s, bwd {tc2 - tc1:.2f} s; {ncpu} OpenMP threads in the blend "
                                             f"stages, 1 in the per-Gaussian and sort stages"}}


if __name__ == "__main__":
    main()
