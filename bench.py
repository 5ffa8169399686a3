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
inside the timed region by the library's stage timers) and `cpu_baseline` (the C oracle timed on the host
cores for one iteration of the same workload; N=1 only).
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


def stage_bytes(P, R, N, Tn, Kbar):
    """Algorithmic bytes per launch of each stage, SURVEY.md 8d (terms of B_fwd / B_bwd regrouped by the
    library's stages; the sort term is the reference-algorithm figure R*24*ceil(bits/8) as 8d prescribes)."""
    sort_passes = 6 if Tn > 4096 else 5  # ceil((32 + msb(Tn)) / 8) for the tile counts used here
    return {
        "preprocess_fwd": P * 48 + P * 63,            # geometry kernel (on the critical path)
        "sh_color_overlapped": P * 12 * Kbar + P * 12,  # SH -> RGB kernel, runs underneath the sorts (side stream)
        "depth_sort_scan": P * 8,
        "tile_binning": P * 20 + R * 12 + R * 24 * sort_passes + R * 8 + Tn * 8,
        "blend_fwd": R * 40 + N * 20,
        "blend_bwd": R * 40 + N * 20 + R * 36,
        "preprocess_bwd": P * 300 + P * 92 + P * (175 + 24 * Kbar),
    }


# kernels that make up each stage (names as rocprofv3 reports them, without arguments) and launches per stage
STAGE_KERNELS = {
    "preprocess_fwd": [("r3::preprocess_geom_kernel", 1)],
    "sh_color_overlapped": [("r3::preprocess_color_kernel<false>", 1)],
    "depth_sort_scan": [("r3::header_reduce_kernel", 1), ("r3::depth_hist_kernel", 1), ("r3::depth_colscan_kernel", 1),
                        ("r3::depth_scatter_kernel", 1), ("r3::depth_bucket_sort_kernel", 1)],
    "tile_binning": [("r3::emit_pairs_kernel", 1), ("r3::radix_digit_scan_kernel", 2), ("r3::radix_scatter_kernel", 2),
                     ("r3::radix_hist_kernel", 1), ("r3::tile_ranges_kernel", 1)],
    "blend_fwd": [("r3::blend_fwd_kernel<2, false>", 1)],
    "blend_bwd": [("r3::blend_bwd_kernel<4>", 1), ("r3::pair_reduce_kernel", 1)],
    "preprocess_bwd": [("r3::preprocess_bwd_kernel", 1)],
}


def pmc_traffic(stage, workload):
    """HBM bytes per launch of the stage's own kernels from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_summary.json; FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes,
    unit KiB).  gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-B read requests as 64 B for wide
    (16 B/lane) loads, so it is doubled; WRITE_SIZE is taken as reported.  None if no committed counters match."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if workload != "metric_500k_1600x1062" or stage not in STAGE_KERNELS or not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    total = 0.0
    for k, launches in STAGE_KERNELS[stage]:
        if k not in pmc or "FETCH_SIZE" not in pmc[k] or "WRITE_SIZE" not in pmc[k]:
            return None
        total += launches * (2.0 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024.0
    return int(total)


def hold_cpu_awake():
    """Linux PM-QoS: keeping /dev/cpu_dma_latency open with value 0 forbids deep CPU C-states while the bench
    runs.  On an otherwise idle many-core host every host-thread wake-up (autograd worker hand-off, the
    num_rendered read-back) otherwise pays a C-state exit; one visit of the GPU box measured 7.2 ms/step with
    1.16 ms of GPU work per step.  Returns the open file (keep a reference) or None when unavailable."""
    try:
        import stat
        import struct
        if not stat.S_ISCHR(os.stat("/dev/cpu_dma_latency").st_mode):
            return None
        f = open("/dev/cpu_dma_latency", "wb", buffering=0)
        f.write(struct.pack("i", 0))
        return f
    except Exception:
        return None


def raise_host_priority(world):
    """The step has one structural host round trip (num_rendered sizes the binning blob) and the host thread spins
    on an event for it.  On a shared box a normal-priority spinning thread can be preempted for whole scheduler
    quanta: two visits showed 2.2 ms/step (447 it/s) with the usual 0.98 ms of GPU stage time per step.  As root,
    single-process runs move the main thread to SCHED_FIFO (the GPU signals the event, no host thread is needed
    for progress); otherwise / additionally the nice value is lowered.  Returns what took effect."""
    took = []
    if world == 1:
        try:
            os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(1))
            took.append("SCHED_FIFO")
        except Exception:
            pass
    if not took:
        try:
            os.nice(-10)
            took.append("nice-10")
        except Exception:
            pass
    return "+".join(took) or None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="metric_500k_1600x1062", choices=list(ss.WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cameras", type=int, default=8)
    ap.add_argument("--host-diag", action="store_true", help="add host-side per-step timing percentiles")
    args = ap.parse_args()
    pm_qos = hold_cpu_awake()  # noqa: F841  (kept open for the lifetime of the process)
    host_prio = None if os.environ.get("R3DGS_BENCH_NO_PRIO") == "1" else raise_host_priority(
        int(os.environ.get("WORLD_SIZE", "1")))
    # run autograd's backward in the calling thread: no hand-off to a per-device worker thread per iteration
    # (same reason as above: a parked thread's wake-up can cost more than the 1.2 ms step)
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
    cam0 = ss.make_camera(W, H, w["f"], None)
    g = ss.make_gaussians(P, cam0, seed=0, degree_mode=w["degree_mode"])
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
    exch = ViewParallelExchange({"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)},
                                P, device) if world > 1 else None
    if exch is not None and os.environ.get("R3DGS_BENCH_NO_ARENA") != "1":
        _C.set_gradient_arena(exch.arena)   # parameter gradients are written straight into the exchange buffer

    born_in_buffer = [None]

    def cam_index(step):
        return (step * world + rank) % len(cams)

    def train_step(step):
        for t in leaves.values():
            t.grad = None
        means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0  # gaussian_renderer/__init__.py:27
        means2D.retain_grad()
        color, radii = dgr.rasterize_gaussians(leaves["means3D"], means2D, leaves["sh"], degrees, empty,
                                               leaves["opacity"], leaves["scales"], leaves["rotations"], empty,
                                               settings[cam_index(step)], 0.0)
        color.backward(dl)
        if exch is not None:
            grads = {k: v.grad for k, v in leaves.items()}
            if step == 0:   # reported once: did autograd keep the arena views as .grad (zero-copy pack)?
                born_in_buffer[0] = sum(exch.arena(k, tuple(g_.shape)) is not None and
                                        g_.data_ptr() == exch.arena(k, tuple(g_.shape)).data_ptr()
                                        for k, g_ in grads.items())
            exch.pack(grads, means2D.grad, radii)
            exch.exchange()
        return radii

    # num_rendered per camera (property of the input; every per-pair byte term scales with it)
    Rs, Vs = [], []
    with torch.no_grad():
        for s_ in settings:
            out = _C.rasterize_gaussians(s_.bg, leaves["means3D"], empty, leaves["opacity"], leaves["scales"],
                                         leaves["rotations"], 1.0, empty, s_.viewmatrix, s_.projmatrix, s_.tanfovx,
                                         s_.tanfovy, H, W, leaves["sh"], degrees, s_.campos, False, False)
            Rs.append(out[0])
            Vs.append(int((out[2] > 0).sum()))

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        train_step(i)
    torch.cuda.synchronize()
    # Timed region: HIP events around the DOMINANT stage only (two records per step).  Timing all seven stages
    # puts 14 more packets on the stream per step, which costs a few percent of a ~1 ms iteration; the full
    # per-stage breakdown is taken in a second, untimed instrumented pass below.
    dom_stage = os.environ.get("R3DGS_BENCH_DOM_STAGE", "blend_bwd")
    prof_mode = os.environ.get("R3DGS_BENCH_PROFILE", "dominant")  # dominant | all | off  (A/B of the timer cost)
    if prof_mode == "all":
        _C.profile_enable(True)
    elif prof_mode == "dominant":
        _C.profile_enable(True, only=[dom_stage])
    _C.profile_read()
    barrier()
    torch.cuda.synchronize()
    host_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        th = time.perf_counter()
        train_step(args.warmup + i)
        host_ms.append(1e3 * (time.perf_counter() - th))
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof_timed = _C.profile_read()
    _C.profile_enable(False)
    # second, untimed pass with every stage timer on: the per-stage breakdown
    _C.profile_enable(True)
    for i in range(min(args.steps, 20)):
        train_step(args.warmup + i)
    torch.cuda.synchronize()
    prof = _C.profile_read()
    _C.profile_enable(False)
    if prof_timed.get(dom_stage, (0, 0))[1]:
        prof[dom_stage] = prof_timed[dom_stage]  # the roofline kernel's time is the one from the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # render-only throughput (render.py's FPS path: forward under no_grad), after the timed region
    with torch.no_grad():
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for i in range(args.steps):
            s_ = settings[cam_index(i)]
            _C.rasterize_gaussians(s_.bg, leaves["means3D"], empty, leaves["opacity"], leaves["scales"],
                                   leaves["rotations"], 1.0, empty, s_.viewmatrix, s_.projmatrix, s_.tanfovx,
                                   s_.tanfovy, H, W, leaves["sh"], degrees, s_.campos, False, False)
        torch.cuda.synchronize()
        render_s = time.perf_counter() - tr0

    used = [cam_index(args.warmup + i) for i in range(args.steps)]
    R_mean = float(np.mean([Rs[k] for k in used]))
    V_mean = float(np.mean([Vs[k] for k in used]))
    sb = stage_bytes(P, R_mean, N, Tn, Kbar)
    stages = {}
    for name, (ms, cnt) in prof.items():
        if cnt:
            avg_ms = ms / cnt
            stages[name] = {"avg_ms": round(avg_ms, 4), "launches": cnt, "alg_bytes": int(sb[name]),
                            "GBps": round(sb[name] / (avg_ms * 1e-3) / 1e9, 1)}
    dom = max((k for k in stages if k != "sh_color_overlapped"), key=lambda k: stages[k]["avg_ms"]) if stages else None
    roofline = None
    if dom:
        A = stages[dom]["GBps"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": A, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(A / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.workload)}
    iters_per_s = args.steps * world / elapsed
    B_iter = P * (718 + 36 * Kbar) + R_mean * 280 + N * 40
    result = {
        "metric": "train iters/s (fwd+bwd) + Mpix/s render, 500k Gaussians @1600x1062",
        "value": round(iters_per_s, 2), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "gaussians": P, "width": W, "height": H, "sh_degree": 3,
                   "views_per_step": world, "visible_mean": round(V_mean), "num_rendered_mean": round(R_mean),
                   "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU",
                   "exchange": "RCCL reduce-scatter+all-gather of 59 fp32 grads + 2 stats / Gaussian, MAX radii"
                   if world > 1 else None,
                   "grads_born_in_exchange_buffer": born_in_buffer[0]},
        "render_mpix_per_s": round(args.steps * N / render_s / 1e6, 1),
        "render_fps": round(args.steps / render_s, 1),
        "roofline": roofline,
        "iter_roofline": {"B_iter_bytes": int(B_iter), "achieved_GBps": round(B_iter * iters_per_s / world / 1e9, 1),
                          "frac_of_8TBps": round(B_iter * iters_per_s / world / 8e12, 4)},
        "stages": stages,
        "stages_note": f"{dom_stage}: HIP events inside the timed region; other stages: separate instrumented pass",
        "host": {"cpu_dma_latency_held": pm_qos is not None, "cpus": os.cpu_count(), "priority": host_prio,
                 "loadavg": [round(x, 1) for x in os.getloadavg()],
                 "host_ms_per_step_min_med_max": [round(sorted(host_ms)[0], 3), round(sorted(host_ms)[len(host_ms) // 2], 3),
                                                  round(sorted(host_ms)[-1], 3)]},
    }
    if args.host_diag:
        hs = sorted(host_ms)
        result["host"].update({"loadavg": os.getloadavg(), "affinity": len(os.sched_getaffinity(0)),
                               "host_ms_per_step_min_med_max": [round(hs[0], 3), round(hs[len(hs) // 2], 3),
                                                                round(hs[-1], 3)]})

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        c = cams[0]
        dl_np = ss.upstream_grad(W, H, seed=1)
        tc0 = time.perf_counter()
        ref = orc.forward(np.zeros(3, np.float32), g["means3D"], None, g["opacity"], g["scales"], g["rotations"], 1.0,
                          None, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, g["sh"],
                          g["degrees"], c.camera_center)
        tc1 = time.perf_counter()
        orc.backward(ref["state"], dl_np, 0.0)
        tc2 = time.perf_counter()
        cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
        result["cpu_baseline"] = {
            "value": round(1.0 / (tc2 - tc0), 4), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"1 full fwd+bwd iteration of the same workload (camera 0) by the C oracle "
                      f"(OpenMP over pixels/tiles in the blend stages, per-Gaussian stages scalar): "
                      f"fwd {tc1 - tc0:.2f} s, bwd {tc2 - tc1:.2f} s"}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
