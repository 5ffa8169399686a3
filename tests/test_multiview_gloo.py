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


def _worker(rank, world, port, two_phase, q, arena=False):
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
    ex.exchange()
    out, gnorm, vis, rmax = ex.unpack()
    q.put((rank, {k: v.clone().numpy() for k, v in out.items()}, gnorm.clone().numpy(), vis.clone().numpy(),
           rmax.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(two_phase, arena=False):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, two_phase, q, arena)) for r in range(world)]
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
