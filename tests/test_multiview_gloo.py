"""world_size-2 `gloo` test of the view-parallel exchange (reduced-3dgs_amd/multiview.py): the N>1 path of
bench.py / training, on CPU.  Each rank holds the gradients of 'its' view; after the exchange both ranks must
hold the SUM of parameter gradients and densification statistics and the MAX of radii."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SHAPES = {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}
P = 257  # deliberately not a multiple of the world size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_inputs(rank):
    g = torch.Generator().manual_seed(100 + rank)
    grads = {k: torch.randn((P,) + s, generator=g) for k, s in SHAPES.items()}
    vgrad = torch.randn(P, 3, generator=g)
    radii = torch.randint(0, 40, (P,), generator=g, dtype=torch.int32)
    radii[torch.rand(P, generator=g) < 0.3] = 0
    return grads, vgrad, radii


def _worker(rank, world, port, two_phase, q, arena=False, use_async=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reduced-3dgs_amd"))
    from multiview import ViewParallelExchange
    ex = ViewParallelExchange(SHAPES, P, torch.device("cpu"), two_phase=two_phase)
    grads, vgrad, radii = _rank_inputs(rank)
    if arena:   # gradients "born" in the exchange buffer, as the rasterizer's backward does with set_gradient_arena
        born = {}
        for k, v in grads.items():
            t = ex.arena(k, tuple(v.shape))
            assert t is not None and t.data_ptr() == ex.flat.data_ptr() + 4 * ex.slices[k][0]
            t.copy_(v)
            born[k] = t
        assert ex.arena("cov3D", (P, 6)) is None and ex.arena("sh", (P, 4, 3)) is None
        grads = born
        sentinel = ex.flat.clone()
    ex.pack(grads, vgrad, radii)
    if arena:   # pack() had nothing to copy for the parameter gradients
        assert torch.equal(ex.flat[:ex.stat_off], sentinel[:ex.stat_off])
    if use_async:   # double-buffered form: the next step's buffer becomes current while this one is in flight
        k = ex.exchange_async()
        assert ex.cur != k and ex.arena("means3D", (P, 3)).data_ptr() != ex.flats[k].data_ptr()
        ex.flat.fill_(7.0)          # "next step" scribbling over the other buffer must not disturb the one in flight
        ex.wait(k)
        out, gnorm, vis, rmax = ex.unpack(k)
    else:
        ex.exchange()
        out, gnorm, vis, rmax = ex.unpack()
    q.put((rank, {k: v.clone().numpy() for k, v in out.items()}, gnorm.clone().numpy(), vis.clone().numpy(),
           rmax.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(two_phase, arena=False, use_async=False):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, two_phase, q, arena, use_async)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ins = [_rank_inputs(r) for r in range(world)]
    for rank, out, gnorm, vis, rmax in results:
        for k in SHAPES:
            np.testing.assert_allclose(out[k], sum(i[0][k] for i in ins).numpy(), rtol=1e-6, atol=1e-6)
        exp_norm = sum((torch.norm(i[1][:, :2], dim=-1) * (i[2] > 0)) for i in ins).numpy()
        np.testing.assert_allclose(gnorm, exp_norm, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(vis, sum((i[2] > 0).float() for i in ins).numpy())
        np.testing.assert_array_equal(rmax, torch.maximum(ins[0][2], ins[1][2]).numpy())


def test_exchange_reduce_scatter_all_gather():
    _run(two_phase=True)


def test_exchange_single_all_reduce():
    _run(two_phase=False)


def test_exchange_with_gradient_arena():
    _run(two_phase=True, arena=True)


def test_exchange_async_double_buffered():
    _run(two_phase=True, arena=True, use_async=True)


# ---- world size 4, Gaussian count not divisible by 4 and changing between steps ---------------------------------------
def _rank_inputs_p(rank, Pn, step):
    g = torch.Generator().manual_seed(1000 * step + 10 * Pn + rank)
    grads = {k: torch.randn((Pn,) + s, generator=g) for k, s in SHAPES.items()}
    vgrad = torch.randn(Pn, 3, generator=g)
    radii = torch.randint(0, 40, (Pn,), generator=g, dtype=torch.int32)
    radii[torch.rand(Pn, generator=g) < 0.3] = 0
    return grads, vgrad, radii


W4_SIZES = [257, 301, 301, 94]   # densify, (same), prune: none divisible by 4


def _w4_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reduced-3dgs_amd"))
    from multiview import ViewParallelExchange
    ex = ViewParallelExchange(SHAPES, W4_SIZES[0], torch.device("cpu"), two_phase=True)
    res = []
    for step, Pn in enumerate(W4_SIZES):
        ex.resize(Pn)                       # what a trainer calls after densify_and_prune
        assert ex.P == Pn and ex.flat.numel() % world == 0
        grads, vgrad, radii = _rank_inputs_p(rank, Pn, step)
        born = {}
        for k, v in grads.items():          # gradients born in the (possibly re-allocated) buffer
            t = ex.arena(k, tuple(v.shape))
            assert t is not None
            t.copy_(v)
            born[k] = t
        ex.pack(born, vgrad, radii)
        k = ex.exchange_async()
        ex.wait(k)
        out, gnorm, vis, rmax = ex.unpack(k)
        res.append(({n: v.clone().numpy() for n, v in out.items()}, gnorm.clone().numpy(), vis.clone().numpy(),
                    rmax.clone().numpy()))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_exchange_with_changing_gaussian_count():
    """Four ranks, P not divisible by the world size (the flat buffer is padded to a multiple of it), P changing between
    steps as densification / pruning does (buffers re-laid out by resize()): every rank ends every step with the SUM of
    the gradients / statistics and the MAX of the radii, and all four replicas hold the same bits."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_w4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step, Pn in enumerate(W4_SIZES):
        ins = [_rank_inputs_p(r, Pn, step) for r in range(world)]
        for rank in range(world):
            out, gnorm, vis, rmax = results[rank][step]
            for k in SHAPES:
                np.testing.assert_allclose(out[k], sum(i[0][k] for i in ins).numpy(), rtol=1e-6, atol=1e-5)
                np.testing.assert_array_equal(out[k], results[0][step][0][k])      # replicas bit-identical
            exp_norm = sum((torch.norm(i[1][:, :2], dim=-1) * (i[2] > 0)) for i in ins).numpy()
            np.testing.assert_allclose(gnorm, exp_norm, rtol=1e-6, atol=1e-5)
            np.testing.assert_array_equal(vis, sum((i[2] > 0).float() for i in ins).numpy())
            want = ins[0][2]
            for i in ins[1:]:
                want = torch.maximum(want, i[2])
            np.testing.assert_array_equal(rmax, want.numpy())


# ---- camera-sharded statistics (SURVEY.md 8e tier 2) ------------------------------------------------------------------
def _cv_partial(cams):
    """numpy model of the per-camera recurrence of calculate_colours_variance (reduced_3dgs.cu:154-198, with the
    reference's mean_old aliasing) over the given cameras: cams = list of (w[P], colour[P,3], dist[P,D])."""
    Pn, D = cams[0][1].shape[0], cams[0][2].shape[1]
    acc, wsum = np.zeros((Pn, D), np.float64), np.zeros((Pn, 1), np.float64)
    mean, S = np.zeros((Pn, 1, 3), np.float64), np.zeros((Pn, 1, 3), np.float64)
    for w, col, dist_ in cams:
        w = w.reshape(-1, 1)
        wsum = wsum + w
        acc = acc + w * dist_
        coeff = np.where(wsum > 0, w / np.maximum(wsum, 1e-300), 0.0).reshape(-1, 1, 1)
        mean = mean + coeff * (col.reshape(-1, 1, 3) - mean)
        S = S + w.reshape(-1, 1, 1) * (col.reshape(-1, 1, 3) - mean) ** 2
    return acc, wsum, mean, S


def _cv_cameras(seed, n):
    rng = np.random.default_rng(seed)
    return [(rng.uniform(0.05, 1.0, 300), rng.uniform(0, 1, (300, 3)), rng.uniform(0, 0.3, (300, 3))) for _ in range(n)]


def _tier2_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reduced-3dgs_amd"))
    from multiview import merge_colour_variance, min_over_ranks
    cams = _cv_cameras(5, 12)
    mine = cams[rank::world]                       # cameras sharded round-robin over the ranks
    part = tuple(torch.from_numpy(x).float() for x in _cv_partial(mine))
    dists, var, mean = merge_colour_variance(part)
    rng = np.random.default_rng(40 + rank)
    px = torch.from_numpy(rng.uniform(0.5, 3.0, 300).astype(np.float32))
    px[rng.random(300) < 0.3] = 10000.0            # unseen by this rank's cameras
    mn = min_over_ranks(px.clone())
    q.put((rank, dists.numpy(), var.numpy(), mean.numpy(), px.numpy(), mn.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_camera_sharded_colour_variance_merge_and_pixel_size_min():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tier2_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cams = _cv_cameras(5, 12)
    acc, wsum, mean, S = _cv_partial(cams)          # all cameras on one rank, in order
    # exact references: weighted mean and the un-aliased weighted variance
    W = sum(c[0] for c in cams).reshape(-1, 1)
    mu = sum(c[0].reshape(-1, 1) * c[1] for c in cams) / W
    M2 = sum(c[0].reshape(-1, 1) * (c[1] - mu) ** 2 for c in cams)
    for rank, dists, var, mean_m, px, mn in results:
        np.testing.assert_allclose(dists, acc / wsum, rtol=1e-5, atol=1e-6)            # plain sums
        np.testing.assert_allclose(mean_m[:, 0], mu, rtol=1e-5, atol=1e-6)               # weighted mean: exact merge
        np.testing.assert_allclose(mean_m, mean, rtol=1e-5, atol=1e-6)
        # variance: the merge is exact on the textbook recurrence; with the reference's aliasing both the sequential
        # and the sharded numbers sit a little below the true weighted variance, by an order-dependent amount
        seq = (S / wsum.reshape(-1, 1, 1))[:, 0]
        true = M2 / W
        assert np.all(var[:, 0] <= true * (1 + 1e-5) + 1e-7)
        assert np.abs(var[:, 0] - seq).max() <= 0.25 * true.max()
        assert np.abs(var[:, 0] - true).mean() <= np.abs(seq - true).mean() * 3 + 1e-6
    np.testing.assert_array_equal(results[0][5], results[1][5])                          # replicas agree bit for bit
    np.testing.assert_array_equal(results[0][5], np.minimum(results[0][4], results[1][4]))
    np.testing.assert_array_equal(results[0][2], results[1][2])


# ---- compact transports (round 6): visible-union rows, bfloat16 SH bands ------------------------------------------------
def _contract_inputs(rank, Pn, step, seen=0.45):
    """Per-rank gradients as the rasterizer's backward leaves them: a Gaussian this view culled (radii == 0) has EXACT zeros
    in every gradient tensor (include/r3dgs_rasterizer.h: every element is written, zeros included).  About half the
    Gaussians are seen per rank, some by no rank at all."""
    g = torch.Generator().manual_seed(7000 * step + 13 * Pn + rank)
    radii = torch.randint(1, 40, (Pn,), generator=g, dtype=torch.int32)
    never = torch.rand(Pn, generator=torch.Generator().manual_seed(99 + step)) < 0.2     # the same rows on every rank
    radii[(torch.rand(Pn, generator=g) > seen) | never] = 0
    vis = radii > 0
    grads = {k: torch.randn((Pn,) + s, generator=g) * vis.view((-1,) + (1,) * len(s)) for k, s in SHAPES.items()}
    # ... and exact zeros in the SH bands above its degree (the degrees are replicated state: the same on every rank)
    deg = _degrees(Pn, step)
    keep = (torch.arange(16)[None, :] < ((deg.view(-1, 1) + 1) ** 2)).unsqueeze(-1)
    grads["sh"] = grads["sh"] * keep
    vgrad = torch.randn(Pn, 3, generator=g) * vis.view(-1, 1)
    return grads, vgrad, radii


def _degrees(Pn, step):
    return torch.randint(0, 4, (Pn,), generator=torch.Generator().manual_seed(555 + step), dtype=torch.int32)


XC_SIZES = [257, 301, 94]   # none divisible by 4; the count changes between steps (resize)


def _xc_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reduced-3dgs_amd"))
    from multiview import ViewParallelExchange
    cpu = torch.device("cpu")
    forms = {"dense": dict(), "sparse": dict(sparse=True), "auto_low": dict(sparse="auto", sparse_threshold=0.05),
             "auto_high": dict(sparse="auto", sparse_threshold=0.999), "bf16": dict(sh_rest_bf16=True),
             "sparse_bf16": dict(sparse=True, sh_rest_bf16=True),
             # SH bands by degree (set_degrees): lossless, alone and on top of the other two
             "bands": dict(), "sparse_bands": dict(sparse=True), "sparse_bands_bf16": dict(sparse=True, sh_rest_bf16=True)}
    exs = {n: ViewParallelExchange(SHAPES, XC_SIZES[0], cpu, two_phase=True, **kw) for n, kw in forms.items()}
    res = []
    for step, Pn in enumerate(XC_SIZES):
        grads, vgrad, radii = _contract_inputs(rank, Pn, step)
        per_form = {}
        for n, ex in exs.items():
            ex.resize(Pn)
            if "bands" in n:
                ex.set_degrees(_degrees(Pn, step).view(-1, 1))
            born = {}
            for k, v in grads.items():      # gradients born in the dense arena, as set_gradient_arena arranges
                t = ex.arena(k, tuple(v.shape))
                assert t is not None
                t.copy_(v)
                born[k] = t
            ex.pack(born, vgrad, radii)
            if step % 2:
                kk = ex.exchange_async()
                ex.wait(kk)
                out, gnorm, vis, rmax = ex.unpack(kk)
            else:
                ex.exchange()
                out, gnorm, vis, rmax = ex.unpack()
            per_form[n] = ({k: v.clone().numpy() for k, v in out.items()}, gnorm.clone().numpy(), vis.clone().numpy(),
                           rmax.clone().numpy(), dict(ex.last), ex.bytes_per_rank(ex.last["rows"]), ex.bytes_per_rank())
        res.append(per_form)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _run_xc(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step, Pn in enumerate(XC_SIZES):
        ins = [_contract_inputs(r, Pn, step) for r in range(world)]
        union = sum((i[2] > 0).int() for i in ins) > 0
        U = int(union.sum())
        assert 0 < U < Pn
        dense0 = results[0][step]["dense"]
        for rank in range(world):
            forms = results[rank][step]
            # (1) the fp32 compact forms are the dense form BIT FOR BIT: the same rank-order sums per element, and rows
            #     outside the union are the exact zeros every rank already holds
            for n in ("sparse", "auto_low", "auto_high", "bands", "sparse_bands"):
                for k in SHAPES:
                    np.testing.assert_array_equal(forms[n][0][k], forms["dense"][0][k], err_msg=f"{n} {k} step {step}")
                for j in (1, 2, 3):
                    np.testing.assert_array_equal(forms[n][j], forms["dense"][j])
            # (2) every form: all replicas hold the same bits
            for n in forms:
                for k in SHAPES:
                    np.testing.assert_array_equal(forms[n][0][k], results[0][step][n][0][k])
                for j in (1, 2, 3):
                    np.testing.assert_array_equal(forms[n][j], results[0][step][n][j])
            # (3) the dense form is the sum / max it always was
            for k in SHAPES:
                np.testing.assert_allclose(forms["dense"][0][k], sum(i[0][k] for i in ins).numpy(), rtol=1e-6, atol=1e-5)
            want = ins[0][2]
            for i in ins[1:]:
                want = torch.maximum(want, i[2])
            np.testing.assert_array_equal(forms["dense"][3], want.numpy())
            # (4) bfloat16 SH bands >= 1: every other tensor, the DC band, statistics and radii are the dense bits; the
            #     higher bands are the fp32 sum of the bf16-rounded per-rank values, rounded once more: within 2^-7 x world x
            #     the largest addend per element (two roundings of relative 2^-8 each -- bfloat16 carries 8 significant bits)
            for n in ("bf16", "sparse_bf16", "sparse_bands_bf16"):
                for k in SHAPES:
                    if k != "sh":
                        np.testing.assert_array_equal(forms[n][0][k], dense0[0][k])
                for j in (1, 2, 3):
                    np.testing.assert_array_equal(forms[n][j], dense0[j])
                np.testing.assert_array_equal(forms[n][0]["sh"][:, 0, :], dense0[0]["sh"][:, 0, :])
                hi, hi_d = forms[n][0]["sh"][:, 1:, :], dense0[0]["sh"][:, 1:, :]
                bound = 2.0 ** -7 * world * np.max([np.abs(i[0]["sh"][:, 1:, :].numpy()) for i in ins], axis=0) + 1e-30
                assert np.all(np.abs(hi - hi_d) <= bound)
                assert np.abs(hi - hi_d).max() > 0          # (it really is the reduced-precision path)
                as_bf16 = torch.from_numpy(hi.copy()).to(torch.bfloat16).to(torch.float32).numpy()
                np.testing.assert_array_equal(as_bf16, hi)   # the replicas continue with bfloat16-representable sums
                np.testing.assert_array_equal(forms["sparse_bf16"][0]["sh"], forms["bf16"][0]["sh"])
                np.testing.assert_array_equal(forms["sparse_bands_bf16"][0]["sh"], forms["bf16"][0]["sh"])
            # (5) what went over the links
            assert forms["sparse"][4]["rows"] == U and forms["sparse"][4]["form"].startswith("visible-union")
            assert forms["auto_high"][4]["rows"] == U                     # union below the threshold: sparse
            assert forms["auto_low"][4]["form"].startswith("dense")       # union above it: the dense form for this step
            assert forms["sparse"][5] < forms["dense"][6] and forms["sparse_bf16"][5] < forms["sparse"][5]
            assert forms["bf16"][6] < 0.7 * forms["dense"][6]             # 248 -> 158 bytes per Gaussian
            # SH bands by degree at a uniform mix of degrees 0..3: 48 -> ~22.5 SH floats per row, 248 -> ~146 bytes
            assert forms["bands"][4]["form"] == "all rows, SH bands by degree" and 0.5 < forms["bands"][5] / forms["dense"][6] < 0.68
            assert forms["sparse_bands"][5] < forms["sparse"][5] and forms["sparse_bands_bf16"][5] < forms["sparse_bands"][5]


def test_world2_compact_transports_equal_dense():
    _run_xc(2)


def test_world4_compact_transports_equal_dense_with_changing_gaussian_count():
    _run_xc(4)
