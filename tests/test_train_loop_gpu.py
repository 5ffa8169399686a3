"""GPU test (-m gpu): an OWN training loop (not the reference's Python) that stresses the things a real run does to the
asynchronous forward -- many cameras whose (tile, Gaussian) pair counts span more than 3x, a Gaussian count that changes
by a densify / prune stand-in every 10 steps, 300 optimisation steps -- and checks, at every step, that the pass the
optimiser consumes (default settings: raster_settings.debug=False, strict mode) is the exact-size path's bit for bit:
image, radii and every gradient.  What the reference guarantees by construction (it sizes its pair buffer from the exact
count, rasterizer_impl.cu:441-450; train.py:63-155 is the schedule mimicked here) has to hold on the reserved path too:
a pass that overflowed its reservation is detected before its outputs are returned and redone.
"""
import numpy as np
import pytest
import torch

import synth_scene as ss

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cameras(W, H, f, n):
    """n cameras whose pair counts differ by far more than the reservation's slack: every fourth one looks straight at
    the scene, the others are turned away by 50-70 degrees and see a third to a half of it (pair counts ~1.1 M vs
    0.36-0.57 M on the scene below, checked with the oracle).  Small rigid perturbations on top."""
    cams = []
    for k in range(n):
        rng = np.random.default_rng(k)
        yaw = [0.0, 0.9, 1.05, 1.2][k % 4] + rng.uniform(-0.03, 0.03)
        cams.append(ss.Camera(W, H, f, f, ss.rot_xyz(rng.uniform(-0.05, 0.05), yaw, rng.uniform(-0.05, 0.05)),
                              rng.uniform(-0.1, 0.1, 3)))
    return cams


def _render(dgr, settings, p, degrees, lam):
    means2D = torch.zeros_like(p["xyz"], requires_grad=True) + 0
    if means2D.requires_grad:   # not under no_grad
        means2D.retain_grad()
    color, radii = dgr.GaussianRasterizer(settings)(
        means3D=p["xyz"], means2D=means2D, shs=p["sh"], degrees=degrees, colors_precomp=None, opacities=p["opacity"],
        scales=torch.exp(p["log_scale"]), rotations=torch.nn.functional.normalize(p["rot"]), cov3D_precomp=None,
        lambda_sh_sparsity=lam)
    return color, radii, means2D


def test_own_loop_every_consumed_pass_equals_the_exact_path():
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    assert _C.is_strict(), "strict mode is the default"
    W, H, P0, f = 800, 560, 150_000, 380.0
    n_cams, steps = 16, 300
    cams = _cameras(W, H, f, n_cams)
    base = ss.make_camera(W, H, f, None)
    g = ss.make_gaussians(P0, base, seed=77, degree_mode="mixed", scale_mu=0.03, zmin=2.0, zmax=8.0)
    bg = _dev(np.array([0.1, 0.2, 0.3], np.float32))

    def settings(c, debug):
        return dgr.GaussianRasterizationSettings(H, W, c.tanfovx, c.tanfovy, bg, 1.0, c._vm, c._pm, 3, c._cp, False, debug)
    for c in cams:   # persistent per-camera tensors, as scene/cameras.py keeps them: the view matrix's address is the key
        c._vm, c._pm, c._cp = _dev(c.world_view_transform), _dev(c.full_proj_transform), _dev(c.camera_center)

    params = {"xyz": _dev(g["means3D"]), "sh": _dev(g["sh"]), "opacity": _dev(g["opacity"]),
              "log_scale": torch.log(_dev(g["scales"])), "rot": _dev(g["rotations"])}
    degrees = _dev(g["degrees"])
    gen = torch.Generator(device="cuda").manual_seed(5)
    targets = {}
    with torch.no_grad():
        for k, c in enumerate(cams):
            targets[k] = _render(dgr, settings(c, True), params, degrees, 0.0)[0].clone()
    params = {k: v.clone().requires_grad_() for k, v in params.items()}
    lrs = {"xyz": 1e-3, "sh": 5e-3, "opacity": 2e-2, "log_scale": 3e-3, "rot": 2e-3}

    def make_opt():
        return torch.optim.Adam([{"params": [params[k]], "lr": lrs[k]} for k in params], eps=1e-15)
    opt = make_opt()

    # pair counts of the cameras at the start: the spread this test is about
    R0 = []
    with torch.no_grad():
        for c in cams:
            out = _C._forward_common(None, bg, params["xyz"], torch.Tensor([]), params["opacity"],
                                     torch.exp(params["log_scale"]), torch.nn.functional.normalize(params["rot"]), 1.0,
                                     torch.Tensor([]), c._vm, c._pm, c.tanfovx, c.tanfovy, H, W, params["sh"], degrees,
                                     c._cp, False, False, exact=True)
            R0.append(out[0].pairs)
    assert max(R0) >= 3 * min(R0), (min(R0), max(R0))

    # everything so far ran on the exact-size path and taught the library every camera's pair count: start the loop
    # from a clean slate, as a training run does
    torch.cuda.synchronize()
    _C.reserve_forget()
    stats0 = _C.pass_stats()
    order = np.random.default_rng(3).permutation(n_cams)
    # the frontal cameras (largest pair counts) are held back until step 120: the per-size advice has by then only seen
    # the cheaper ones, so the first of them overflows its reservation (1.5 x + 64 k) and must be redone
    late = set(k for k in range(n_cams) if k % 4 == 0)
    assert min(R0[k] for k in late) > 1.5 * max(R0[k] for k in range(n_cams) if k not in late) + 65536
    checked = 0
    for step in range(steps):
        pool = [int(k) for k in order if (int(k) not in late or step >= 120)]
        k = pool[step % len(pool)]
        c = cams[k]
        lam = 0.05 if step % 3 == 0 else 0.0
        # (a) what training consumes: default settings
        opt.zero_grad(set_to_none=True)
        color, radii, m2d = _render(dgr, settings(c, False), params, degrees, lam)
        loss = (color - targets[k]).abs().mean()
        loss.backward()
        got = {n: p.grad.clone() for n, p in params.items()}
        got_m2d = m2d.grad.clone()
        # (b) the same inputs through the exact-size path
        for p in params.values():
            p.grad = None
        color_x, radii_x, m2d_x = _render(dgr, settings(c, True), params, degrees, lam)
        (color_x - targets[k]).abs().mean().backward()
        assert torch.equal(color, color_x), f"step {step} cam {k}: image differs from the exact path"
        assert torch.equal(radii, radii_x), f"step {step}: radii"
        assert torch.equal(got_m2d, m2d_x.grad), f"step {step}: means2D grad"
        for n, p in params.items():
            assert torch.equal(got[n], p.grad), f"step {step} cam {k}: grad {n} differs from the exact path"
            assert torch.isfinite(p.grad).all()
        checked += 1
        opt.step()
        if step % 10 == 9:   # densify / prune stand-in (train.py:132-147): clone 6 % with jitter, drop 3 %, new optimiser state
            with torch.no_grad():
                P = params["xyz"].shape[0]
                idx = torch.randperm(P, generator=gen, device="cuda")
                clone, keep = idx[: P * 6 // 100], idx[P * 3 // 100:]
                new = {}
                for n, p in params.items():
                    extra = p[clone].clone()
                    if n == "xyz":
                        extra += 0.01 * torch.randn(extra.shape, generator=gen, device="cuda")
                    new[n] = torch.cat([p[keep], extra]).contiguous()
                degrees = torch.cat([degrees[keep], degrees[clone]]).contiguous()
            params = {n: v.requires_grad_() for n, v in new.items()}
            opt = make_opt()
    st = _C.pass_stats()
    redone = st["redone_passes"] - stats0["redone_passes"]
    reserved = st["reserved_passes"] - stats0["reserved_passes"]
    print(f"\\nown loop: {checked} steps checked, P {P0} -> {params['xyz'].shape[0]}, pair counts {min(R0)}..{max(R0)}, "
          f"{reserved} reserved passes, {redone} redone on overflow")
    assert reserved >= steps // 2, "the loop did not run on the asynchronous path"
    assert redone >= 1, "no pass overflowed its reservation: the redo path was not exercised"
    assert redone <= steps // 10, "the per-camera advice is not learning"


def test_strict_redo_and_lossy_opt_out():
    """The same overflowing reservation through strict mode (default: result == exact path, one redo counted) and with
    strict mode off (the old behaviour: farthest pairs dropped, pass flagged)."""
    from diff_gaussian_rasterization import _C
    W, H, P = 320, 240, 8000
    cam = ss.make_camera(W, H, 250.0, 6)
    g = ss.make_gaussians(P, cam, seed=41, degree_mode="all3", scale_mu=0.03)
    args = (_dev(np.array([0.3, 0.2, 0.1], np.float32)), _dev(g["means3D"]), torch.Tensor([]), _dev(g["opacity"]),
            _dev(g["scales"]), _dev(g["rotations"]), 1.0, torch.Tensor([]), _dev(cam.world_view_transform),
            _dev(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, H, W, _dev(g["sh"]), _dev(g["degrees"]),
            _dev(cam.camera_center), False, False)
    exact = _C._forward_common(None, *args, exact=True)
    R = exact[0].pairs
    s0 = _C.pass_stats()
    out = _C._forward_common(None, *args, _reserve=R // 2)            # strict (default)
    s1 = _C.pass_stats()
    assert s1["redone_passes"] == s0["redone_passes"] + 1
    assert not out[0].truncated and out[0].pairs == R and int(out[0]) == int(exact[0])
    assert torch.equal(out[1], exact[1]) and torch.equal(out[2], exact[2])
    # counter mode accumulates into its outputs: the redo must not count the truncated pass as well
    ref_c = _C._forward_common(None, *args, counters=(torch.zeros(P, dtype=torch.int32, device="cuda"),
                                                      torch.zeros(P, device="cuda")), exact=True)
    touched, transm = torch.zeros(P, dtype=torch.int32, device="cuda"), torch.zeros(P, device="cuda")
    touched_x, transm_x = torch.zeros_like(touched), torch.zeros_like(transm)
    _C._forward_common(None, *args, counters=(touched_x, transm_x), exact=True)
    _C._forward_common(None, *args, counters=(touched, transm), _reserve=R // 2)
    assert torch.equal(touched, touched_x)
    assert torch.allclose(transm, transm_x, rtol=1e-5, atol=1e-4)   # float atomics: order-dependent low bits
    del ref_c
    lossy = _C._forward_common(None, *args, _reserve=R // 2, _strict_override=False)
    assert lossy[0].truncated and not torch.equal(lossy[1], exact[1])


def test_freed_camera_matrix_does_not_leave_its_pair_count_to_the_next_owner_of_the_address():
    """Per-camera pair counts are kept under the device address of the camera's view matrix (ADVICE r3 / VERDICT r3 weak 15).
    A camera with few pairs is freed and the allocator hands its address to a camera with 4x the pairs.  Without the
    host layer's `r3dgs_reserve_forget_view` on the tensor's death the newcomer would inherit the small count, overflow
    its reservation and be redone (strict mode) on its first pass; with it, the newcomer is unknown and gets the image
    size's largest recent count (the first, large camera's): no redo, and the results are the exact path's."""
    import gc

    from diff_gaussian_rasterization import _C
    W, H, P = 320, 240, 30000
    cam = ss.make_camera(W, H, 250.0, 6)
    g = ss.make_gaussians(P, cam, seed=43, degree_mode="all0", scale_mu=0.2)
    fixed = dict(bg=_dev(np.zeros(3, np.float32)), m=_dev(g["means3D"]), op=_dev(g["opacity"]), sc=_dev(g["scales"]),
                 rot=_dev(g["rotations"]), sh=_dev(g["sh"]), deg=_dev(g["degrees"]), campos=_dev(cam.camera_center))

    def view(zoom):
        """The same camera pulled back by `zoom`: fewer pairs (splats shrink, many leave the rects)."""
        v = np.array(cam.world_view_transform, np.float32, copy=True)
        v[3, 2] += zoom
        proj = np.array(cam.full_proj_transform, np.float32, copy=True)
        p = (v.astype(np.float64) @ np.linalg.inv(np.array(cam.world_view_transform, np.float64)) @ proj.astype(np.float64))
        return v, p.astype(np.float32)

    def fwd(vm_t, pm_t, **kw):
        return _C._forward_common(None, fixed["bg"], fixed["m"], torch.Tensor([]), fixed["op"], fixed["sc"], fixed["rot"], 1.0,
                                  torch.Tensor([]), vm_t, pm_t, cam.tanfovx, cam.tanfovy, H, W, fixed["sh"], fixed["deg"],
                                  fixed["campos"], False, False, **kw)

    _C.reserve_forget()
    v_big, p_big = view(0.0)
    v_small, p_small = view(25.0)
    big1_vm, big1_pm = _dev(v_big), _dev(p_big)
    pairs_big = fwd(big1_vm, big1_pm, exact=True)[0].pairs
    for _ in range(3):
        assert not fwd(big1_vm, big1_pm)[0].truncated
    # one address, two tensor objects one after the other: a view into a pool stands in for "the allocator handed the freed
    # block to the next camera" (the caching allocator's choice of block cannot be forced from a test)
    pool = torch.empty(16, dtype=torch.float32, device="cuda")
    small_vm, small_pm = pool.view(4, 4), _dev(p_small)
    small_vm.copy_(torch.from_numpy(v_small))
    pairs_small = fwd(small_vm, small_pm)[0].pairs
    for _ in range(3):
        fwd(small_vm, small_pm)
    # the small camera's reservation (1.5 x its count + 64 k, rounded up on a 9 % grid) cannot hold the big one
    assert pairs_big > 2 * (1.5 * pairs_small + 65536), (pairs_big, pairs_small)
    torch.cuda.synchronize()
    addr = small_vm.data_ptr()
    del small_vm
    gc.collect()
    big2_vm = pool.view(4, 4)
    assert big2_vm.data_ptr() == addr
    big2_vm.copy_(torch.from_numpy(v_big))
    s0 = _C.pass_stats()
    out = fwd(big2_vm, big1_pm)
    s1 = _C.pass_stats()
    assert s1["redone_passes"] == s0["redone_passes"], "the newcomer inherited the freed camera's pair count"
    assert s1["reserved_passes"] == s0["reserved_passes"] + 1 and out[0].pairs == pairs_big
    ref = fwd(big1_vm, big1_pm, exact=True)
    assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
    # and the counter-experiment: with the lifetime tracking switched off the same sequence does inherit the stale count and
    # pays a redo (strict mode still returns the exact result) -- the test above is not vacuous
    track, _C._track_view = _C._track_view, lambda vm: None
    try:
        _C.reserve_forget()
        fwd(big1_vm, big1_pm, exact=True)
        for _ in range(3):
            fwd(big1_vm, big1_pm)
        v1 = pool.view(4, 4)
        v1.copy_(torch.from_numpy(v_small))
        for _ in range(4):
            fwd(v1, small_pm)
        torch.cuda.synchronize()
        del v1
        v2 = pool.view(4, 4)
        v2.copy_(torch.from_numpy(v_big))
        s0 = _C.pass_stats()
        out2 = fwd(v2, big1_pm)
        assert _C.pass_stats()["redone_passes"] == s0["redone_passes"] + 1
        assert torch.equal(out2[1], ref[1])
    finally:
        _C._track_view = track
        _C.reserve_forget()
