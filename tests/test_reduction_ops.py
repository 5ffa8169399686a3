"""Operators either side of the rasterizer: redundancy score, 1-D k-means codebooks and
nearest neighbours.  CPU tests pin the numpy oracle (oracle/reduction_ops.py) to the reference's own Python
restatement (tests/golden/ref_pixel_size.npz, made by tests/golden/make_golden.py from scene/__init__.py:103-141)
and to hand-checkable cases; the `-m gpu` tests compare the HIP operators with the oracle through the reference's
operator names (`diff_gaussian_rasterization._C.*`, `simple_knn._C.*`).

Bars: integer / index / boolean outputs bit-exact (threshold decisions within 1e-5 of their threshold are excluded
and counted); squared distances bit-exact (same fp32 expression, no contraction); other floats 1e-5 relative."""
import os

import numpy as np
import pytest

import synth_scene as ss
from oracle import reduction_ops as ro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------------------------ CPU: oracle pins
def test_oracle_pixel_size_matches_reference_python():
    d = np.load(os.path.join(GOLD, "ref_pixel_size.npz"))
    got, ambig = ro.min_pixel_size(d["w2ndc"], d["w2ndc_inv"], d["means3D"], d["image_height"], d["image_width"],
                                   want_ambig=True)
    ref = d["pixel_sizes"]
    ok = ~ambig
    assert ok.sum() > 2900
    assert np.array_equal(got[ok] == 10000, ref[ok] == 10000)          # same set of unseen centres
    seen = ok & (ref[:, 0] != 10000)
    assert seen.sum() > 2000
    # the reference's torch version inverts in a different operation order (row-vector matmul, exact 1/w):
    # an fp32 inverse of a projection matrix is only good to ~1e-3 relative
    np.testing.assert_allclose(got[seen], ref[seen], rtol=2e-3)


def test_oracle_intersection_hand_case():
    # query 0 at the origin with radius 0.5; neighbours on the x axis with semi-axes (1, .1, .1):
    # the test is |dx| < 1 + 0.5 for the identity rotation
    means = np.array([[0, 0, 0], [1.4, 0, 0], [1.6, 0, 0], [0, 0.7, 0]], np.float32)
    scales = np.tile(np.array([[1.0, 0.1, 0.1]], np.float32), (4, 1))
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (4, 1))
    nbr = np.array([[1, 2, 3]] * 4, np.int32)
    radius = np.full((4, 1), 0.5, np.float32)
    cnt, mask, _ = ro.sphere_ellipsoid_intersection(means, scales, rot, nbr, radius, 3)
    assert mask[0].tolist() == [True, False, False] and cnt[0, 0] == 1
    # a 90 degree rotation about z of the QUERY (the reference uses R[idx]) swaps the roles of x and y
    rot[0] = [np.sqrt(0.5), 0, 0, np.sqrt(0.5)]
    _, mask, _ = ro.sphere_ellipsoid_intersection(means, scales, rot, nbr, radius, 3)
    assert mask[0].tolist() == [False, False, True]     # (0, .7, 0) now lies along the long axis, x = 1.4 does not
    means[3] = [0, 1.6, 0]
    _, mask, _ = ro.sphere_ellipsoid_intersection(means, scales, rot, nbr, radius, 3)
    assert mask[0].tolist() == [False, False, False]


def test_oracle_min_redundancy_hand_case():
    red = np.array([[5], [2], [9]], np.int32)
    nbr = np.array([[0, 1], [1, 2], [2, 0]], np.int32)
    mask = np.array([[1, 1], [1, 0], [0, 1]], bool)
    out = ro.min_redundancy(red, nbr, mask, 2)
    # 0 <- {5 (from 0), 9 (from 2)}, 1 <- {5, 2}, 2 <- nothing (masked) -> P
    assert out[:, 0].tolist() == [3, 2, 3]     # initial value P=3 caps everything


def test_oracle_kmeans_hand_case():
    v = np.array([0.0, 0.1, 0.2, 10.0, 10.2, 5.0], np.float32)
    c = np.array([0.0, 10.0, 100.0], np.float32)
    ids, centers, it = ro.kmeans(v, c, 1e-4, 50)
    # update 1: 5.0 is equidistant from 0 and 10 -> first index; centre 2 is empty -> NaN -> 0 (reduced_3dgs.cu:321);
    # update 2: that 0 recaptures {0, .1, .2} from centre 0 (now 1.325), which keeps only 5.0
    assert ids[:, 0].tolist() == [2, 2, 2, 1, 1, 0]
    np.testing.assert_allclose(centers, [5.0, 10.1, 0.1], rtol=1e-6)
    _, c1, _ = ro.kmeans(v, c, 0.0, 1)
    assert c1[2] == 0.0 and abs(c1[0] - 1.325) < 1e-6
    # first index wins on ties
    assert ro.kmeans_assign(np.array([1.0], np.float32), np.array([2.0, 0.0, 2.0], np.float32))[0] == 0
    assert it < 50


def test_oracle_knn_matches_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(3)
    p = rng.normal(0, 1, (4000, 3)).astype(np.float32)
    d2, idx = ro.knn_bruteforce(p, 8)
    dd, ii = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=9)
    assert np.array_equal(idx, ii[:, 1:].astype(np.int32))
    np.testing.assert_allclose(d2, dd[:, 1:] ** 2, rtol=1e-5, atol=1e-12)
    m = ro.knn_mean_dist3(p)
    np.testing.assert_allclose(m, (dd[:, 1:4] ** 2).mean(1), rtol=1e-5)
    # fewer points than K: unfilled slots keep the reference's initial values
    d2, idx = ro.knn_bruteforce(p[:3], 4)
    assert (idx[:, 2:] == -1).all() and (d2[:, 2:] == np.finfo(np.float32).max).all()


# ---------------------------------------------------------------------------------------------------- GPU parity
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(P, seed):
    cam = ss.make_camera(400, 300, 350.0)
    return ss.make_gaussians(P, cam, seed=seed, behind_frac=0.05)


@pytest.mark.gpu
def test_gpu_min_pixel_size():
    from diff_gaussian_rasterization import _C
    d = np.load(os.path.join(GOLD, "ref_pixel_size.npz"))
    want, ambig = ro.min_pixel_size(d["w2ndc"], d["w2ndc_inv"], d["means3D"], d["image_height"], d["image_width"],
                                    want_ambig=True)
    got = _C.find_minimum_projected_pixel_size(_dev(d["w2ndc"]), _dev(d["w2ndc_inv"]), _dev(d["means3D"]),
                                               _dev(d["image_height"]), _dev(d["image_width"])).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.float32
    ok = ~ambig
    assert np.array_equal(got[ok] == 10000, want[ok] == 10000)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-5)
    seen = ok & (d["pixel_sizes"][:, 0] != 10000)
    np.testing.assert_allclose(got[seen], d["pixel_sizes"][seen], rtol=2e-3)   # the reference's own torch version
    # no cameras: everything unseen
    none = _C.find_minimum_projected_pixel_size(_dev(d["w2ndc"][:0]), _dev(d["w2ndc_inv"][:0]), _dev(d["means3D"]),
                                                _dev(d["image_height"][:0]), _dev(d["image_width"][:0]))
    assert (none == 10000).all()


@pytest.mark.gpu
@pytest.mark.parametrize("P,knn", [(20000, 30), (5003, 31), (777, 1), (4096, 64), (300, 70)])
def test_gpu_redundancy_score(P, knn):
    """sphere_ellipsoid_intersection + allocate_minimum_redundancy_value as scene/__init__.py:143-174 chains them."""
    import torch
    from diff_gaussian_rasterization import _C
    g = _scene(P, seed=P)
    rng = np.random.default_rng(P)
    d2, nbr = ro.knn_bruteforce(g["means3D"], knn)
    # sphere radius around the median neighbour distance, so that both outcomes occur whatever the density
    radius = (np.sqrt(d2[:, knn // 2:knn // 2 + 1]) * np.exp(rng.normal(0, 0.5, (P, 1)))).astype(np.float32)
    cnt, mask, ambig = ro.sphere_ellipsoid_intersection(g["means3D"], g["scales"], g["rotations"], nbr, radius, knn)
    red, msk = _C.sphere_ellipsoid_intersection(_dev(g["means3D"]), _dev(g["scales"]), _dev(g["rotations"]), _dev(nbr),
                                                _dev(radius), knn)
    assert red.dtype == torch.int32 and msk.dtype == torch.bool and tuple(msk.shape) == (P, knn)
    msk_h, red_h = msk.cpu().numpy(), red.cpu().numpy()
    assert ambig.mean() < 1e-3
    assert np.array_equal(msk_h[~ambig], mask[~ambig])
    assert 0.02 < mask.mean() < 0.98                              # the case exercises both outcomes
    assert np.array_equal(red_h[:, 0], msk_h.sum(1))              # the count is that of the kernel's own mask
    clean = ~ambig.any(1)
    assert np.array_equal(red_h[clean], cnt[clean])

    # second operator on the first one's (device) outputs, plus the self column the caller prepends
    red1 = red + 1
    idx1 = torch.cat((torch.arange(P, device="cuda", dtype=torch.int32).view(-1, 1), _dev(nbr)), dim=1)
    msk1 = torch.cat((torch.ones((P, 1), device="cuda", dtype=torch.bool), msk), dim=1)
    (mn,) = _C.allocate_minimum_redundancy_value(red1, idx1, msk1, knn + 1)
    want = ro.min_redundancy(red1.cpu().numpy(), idx1.cpu().numpy(), msk1.cpu().numpy(), knn + 1)
    assert mn.dtype == torch.int32 and np.array_equal(mn.cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_kmeans_one_update_and_fixed_point():
    import torch
    from diff_gaussian_rasterization import _C
    rng = np.random.default_rng(9)
    n = 200_003
    v = np.concatenate([rng.normal(-2, 0.3, n // 3), rng.normal(0.5, 1.0, n // 3), rng.normal(4, 0.1, n - 2 * (n // 3))])
    v = rng.permutation(v).astype(np.float32)
    c0 = v[rng.integers(0, n, 256)].copy()
    c0[17] = c0[3]                                    # duplicate centres: the later one must end up empty -> 0
    # one update: centres are the means of the clusters of the INITIAL centres, ids are nearest to the NEW centres
    ids, centers, iters = _C.kmeans_cuda(_dev(v).view(-1, 1), _dev(c0), 0.0, 1, _want_iterations=True)
    ids, centers = ids.cpu().numpy(), centers.cpu().numpy()
    assert ids.shape == (n, 1) and ids.dtype == np.int32 and centers.shape == (256,) and int(iters) == 1
    a0 = ro.kmeans_assign(v, c0)
    sums = np.bincount(a0, weights=v.astype(np.float64), minlength=256)
    sizes = np.bincount(a0, minlength=256)
    want = np.where(sizes > 0, sums / np.maximum(sizes, 1), 0.0)
    assert sizes[17] == 0 and centers[17] == 0.0
    np.testing.assert_allclose(centers, want, rtol=2e-5, atol=1e-6)
    assert np.array_equal(ids[:, 0], ro.kmeans_assign(v, centers))       # bit-exact against the returned centres

    # full run: same answer as the oracle's Lloyd loop up to the fp32 summation order
    ids, centers, iters = _C.kmeans_cuda(_dev(v).view(-1, 1), _dev(c0), 1e-4, 60, _want_iterations=True)
    o_ids, o_centers, o_it = ro.kmeans(v, c0, 1e-4, 60)
    centers = centers.cpu().numpy()
    assert abs(int(iters) - o_it) <= 1
    if int(iters) == o_it:
        np.testing.assert_allclose(centers, o_centers, rtol=1e-3, atol=1e-4)
    assert np.array_equal(ids.cpu().numpy()[:, 0], ro.kmeans_assign(v, centers))
    # codebook use (scene/gaussian_model.py:33-44): evaluate() error is bounded by the cluster radius
    recon = centers[ids.cpu().numpy()[:, 0]]
    assert np.abs(recon - v).mean() < 0.02


@pytest.mark.gpu
def test_gpu_kmeans_edge_cases():
    import torch
    from diff_gaussian_rasterization import _C
    v = _dev(np.array([0.0, 0.1, 0.2, 10.0, 10.2, 5.0], np.float32)).view(-1, 1)
    ids, centers = _C.kmeans_cuda(v, _dev(np.array([0.0, 10.0, 100.0], np.float32)), 1e-4, 500)
    o_ids, o_centers, _ = ro.kmeans(v.cpu().numpy(), np.array([0.0, 10.0, 100.0], np.float32), 1e-4, 500)
    assert np.array_equal(ids.cpu().numpy(), o_ids) and np.allclose(centers.cpu().numpy(), o_centers, atol=1e-6)
    # max_iterations = 0: just the assignment to the given centres
    ids, centers = _C.kmeans_cuda(v, _dev(np.array([0.0, 10.0], np.float32)), 1e-4, 0)
    assert ids.view(-1).tolist() == [0, 0, 0, 1, 1, 0] and centers.tolist() == [0.0, 10.0]
    with pytest.raises(RuntimeError):
        _C.kmeans_cuda(v, torch.zeros(2000, device="cuda"), 1e-4, 1)
    with pytest.raises(RuntimeError):
        _C.kmeans_cuda(v.cpu(), torch.zeros(2), 1e-4, 1)                 # no CPU path


@pytest.mark.gpu
def test_gpu_kmeans_tie_semantics():
    """The first-index rule of updateIdsCUDA under exact ties, duplicate / near-duplicate / infinite centres."""
    import torch
    from diff_gaussian_rasterization import _C
    rng = np.random.default_rng(21)
    one = np.float32(1.0000001)
    hand_c = np.array([5, one, np.nextafter(one, np.float32(2)), 5, -3, 0, -0.0, 1e30, np.inf, -np.inf, 2.5, 2.5],
                      np.float32)
    hand_v = np.array([100, -100, 1, 1.0000001, 1.0000002, 3.75, 0, -1.5, 4.99, 1e30, 3e38, -3e38, np.inf, -np.inf,
                       1e-30, 2.5, 1.75], np.float32)
    cases = [(hand_v, hand_c)]
    for k in range(4):                                 # coarse grids: most values are equidistant from two centres
        nc = [7, 256, 300, 1024][k]
        c = (rng.integers(-40, 40, nc) / 4).astype(np.float32)
        v = (rng.integers(-400, 400, 50_000) / 8).astype(np.float32)
        cases.append((v, c))
    c = rng.normal(0, 1, 256).astype(np.float32)
    c[100:110] = c[5] + (np.arange(10) * 1e-7).astype(np.float32)       # a cluster of near-duplicate centres
    cases.append((rng.normal(0, 3, 100_000).astype(np.float32), c))
    for v, c in cases:
        ids, cen = _C.kmeans_cuda(_dev(v).view(-1, 1), _dev(c), 1e-4, 0)
        assert np.array_equal(cen.cpu().numpy(), c, equal_nan=True)
        assert np.array_equal(ids.cpu().numpy()[:, 0], ro.kmeans_assign(v, c))
    # and through updates: every intermediate centre set comes from this rule, compare the whole trajectory
    v, c = cases[2]
    for its in (1, 2, 5):
        ids, cen = _C.kmeans_cuda(_dev(v).view(-1, 1), _dev(c), 0.0, its)
        o_ids, o_cen, _ = ro.kmeans(v, c, 0.0, its)
        np.testing.assert_allclose(cen.cpu().numpy(), o_cen, rtol=1e-6, atol=1e-7)
        assert np.array_equal(ids.cpu().numpy()[:, 0], ro.kmeans_assign(v, cen.cpu().numpy()))


def _check_knn(points, K):
    from simple_knn._C import distIndex2
    P = points.shape[0]
    d, i = distIndex2(_dev(points), K)
    assert tuple(d.shape) == (P * K,) and tuple(i.shape) == (P * K,)
    want_d, want_i = ro.knn_bruteforce(points, K)
    got_d, got_i = d.view(P, K).cpu().numpy(), i.view(P, K).cpu().numpy()
    assert np.array_equal(got_d, want_d)              # same fp32 expression, no contraction: bit-exact
    assert np.array_equal(got_i, want_i)              # ascending by (distance, index)


@pytest.mark.gpu
@pytest.mark.parametrize("P,K", [(20000, 30), (4097, 3), (1000, 64), (130, 100), (64, 1), (5, 8), (1, 2)])
def test_gpu_knn_exact(P, K):
    rng = np.random.default_rng(P + K)
    pts = (rng.normal(0, 1, (P, 3)) * np.array([3.0, 1.0, 0.2])).astype(np.float32)
    _check_knn(pts, K)


def test_oracle_knn_query_is_knn_on_the_full_sets():
    """knn_query_bruteforce with every point as query and candidate is knn_bruteforce."""
    rng = np.random.default_rng(3)
    p = rng.normal(0, 1, (300, 3)).astype(np.float32)
    p[10] = p[11]                                    # a duplicate position: tie decided by index
    allidx = np.arange(300, dtype=np.int32)
    d, i = ro.knn_query_bruteforce(p, allidx, allidx, 7)
    d0, i0 = ro.knn_bruteforce(p, 7)
    assert np.array_equal(d, d0) and np.array_equal(i, i0)


@pytest.mark.gpu
@pytest.mark.parametrize("P,Q,N,K", [(5000, 700, 1800, 12), (3000, 3000, 3000, 5), (513, 40, 3, 8), (2000, 257, 1025, 1),
                                      (100, 10, 0, 4)])
def test_gpu_knn_query_exact(P, Q, N, K):
    """simple_knn._C.distIndexQ (spatial.cu:43-58): bit-exact against the brute force, including duplicate candidate
    indices (a set), queries that are candidates themselves (excluded by index), fewer candidates than K, out-of-range
    indices and an empty candidate list."""
    import torch
    from simple_knn._C import distIndexQ
    rng = np.random.default_rng(P + Q + N + K)
    pts = (rng.normal(0, 1, (P, 3)) * np.array([2.0, 1.0, 0.3])).astype(np.float32)
    pts[5] = pts[6]
    qi = rng.integers(0, P, Q).astype(np.int32)
    ni = rng.integers(0, P, N).astype(np.int32)      # with repetitions
    if N > 10:
        ni[:5] = qi[:5]                              # queries among their own candidates
        ni[7] = -3                                   # ignored
        qi[-1] = P + 9                               # ignored: row stays unfilled
    d, i = distIndexQ(_dev(pts), _dev(qi), _dev(ni), K)
    assert tuple(d.shape) == (Q * K,) and tuple(i.shape) == (Q * K,) and i.dtype == torch.int32
    want_d, want_i = ro.knn_query_bruteforce(pts, qi, ni, K)
    assert np.array_equal(d.view(Q, K).cpu().numpy(), want_d)
    assert np.array_equal(i.view(Q, K).cpu().numpy(), want_i)


@pytest.mark.gpu
def test_gpu_knn_duplicates_and_clusters():
    rng = np.random.default_rng(4)
    base = rng.normal(0, 1, (500, 3)).astype(np.float32)
    pts = np.concatenate([base, base[:200], base[:50], rng.normal(8, 0.01, (900, 3)).astype(np.float32)])
    _check_knn(rng.permutation(pts), 10)              # zero distances and ties: order decided by index
    _check_knn(np.zeros((300, 3), np.float32), 5)     # degenerate bounding box
    line = np.zeros((3000, 3), np.float32)
    line[:, 0] = np.arange(3000)                      # exact ties left/right on a lattice
    _check_knn(line, 4)


@pytest.mark.gpu
def test_gpu_dist_cuda2_and_scene_scale():
    """distCUDA2 (scene/gaussian_model.py:186 initial scales) and a scene-sized distIndex2 against a k-d tree."""
    import torch
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2, distIndex2
    g = _scene(300_000, seed=1)
    pts = g["means3D"]
    m = distCUDA2(_dev(pts)).cpu().numpy()
    d, i = distIndex2(_dev(pts), 30)
    d, i = d.view(-1, 30).cpu().numpy(), i.view(-1, 30).cpu().numpy()
    assert np.array_equal(m, ((d[:, 0] + d[:, 1]) + d[:, 2]) / np.float32(3.0))
    dd, ii = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=31, workers=-1)
    np.testing.assert_allclose(d, dd[:, 1:] ** 2, rtol=2e-5, atol=1e-12)
    assert (i == ii[:, 1:]).mean() > 0.9999           # fp32 vs fp64 near-ties may swap neighbours
    small = ro.knn_mean_dist3(pts[:5000])
    assert np.array_equal(distCUDA2(_dev(pts[:5000])).cpu().numpy(), small)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 3))                 # no CPU path


@pytest.mark.gpu
def test_gpu_pack_view_stats_and_exchange_pack():
    """The fused per-view statistics kernel against the torch expressions it replaces, directly and through
    ViewParallelExchange.pack (world size 1)."""
    import torch
    from diff_gaussian_rasterization import _C
    from multiview import ViewParallelExchange
    P = 10_007
    g = torch.Generator(device="cuda").manual_seed(3)
    vg = torch.randn(P, 3, generator=g, device="cuda")
    radii = torch.randint(0, 50, (P,), generator=g, device="cuda", dtype=torch.int32)
    radii[torch.rand(P, generator=g, device="cuda") < 0.3] = 0
    n, v, r = torch.empty(P, device="cuda"), torch.empty(P, device="cuda"), torch.empty(P, device="cuda", dtype=torch.int32)
    _C.pack_view_stats(vg, radii, n, v, r)
    vis = radii > 0
    want = torch.sqrt(vg[:, 0] * vg[:, 0] + vg[:, 1] * vg[:, 1]) * vis
    assert torch.allclose(n, want, rtol=1e-6, atol=0) and torch.equal(v, vis.float()) and torch.equal(r, radii)
    shapes = {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}
    ex = ViewParallelExchange(shapes, P, torch.device("cuda"))
    grads = {k: torch.randn((P,) + s, generator=g, device="cuda") for k, s in shapes.items()}
    ex.pack(grads, vg, radii)
    out, gnorm, visible, rmax = ex.unpack()
    for k in shapes:
        assert torch.equal(out[k], grads[k])
    assert torch.allclose(gnorm, want, rtol=1e-6, atol=0) and torch.equal(visible, vis.float()) and torch.equal(rmax, radii)


@pytest.mark.gpu
def test_gpu_reduce_shards_sum_and_max():
    """Local half of the view-parallel exchange (r3dgs_reduce_shards): fp32 SUM below sum_len, int32 MAX above, for a
    shard that straddles the boundary -- against torch, bit for bit (same rank order of the additions)."""
    import torch
    from diff_gaussian_rasterization import _C
    world, shard = 8, 1000
    g = torch.Generator().manual_seed(3)
    for begin, sum_len in ((0, 5000), (3 * shard, 3 * shard + 417), (5 * shard, 4000)):
        recv = torch.randn(world, shard, generator=g)
        n_sum = min(max(sum_len - begin, 0), shard)
        ints = torch.randint(0, 500, (world, shard - n_sum), generator=g, dtype=torch.int32)
        recv[:, n_sum:] = ints.view(torch.float32)
        out = torch.empty(shard, device="cuda")
        _C.reduce_shards(recv.cuda().contiguous().view(-1), world, begin, sum_len, out)
        acc = recv[0, :n_sum].clone()
        for w in range(1, world):
            acc += recv[w, :n_sum]
        assert torch.equal(out[:n_sum].cpu(), acc)
        assert torch.equal(out[n_sum:].cpu().view(torch.int32), ints.amax(0))


@pytest.mark.gpu
def test_gpu_reduce_shards_mixed_matches_the_host_path():
    """r3dgs_reduce_shards_mixed: fp32 SUM | bfloat16 pairs (fp32 accumulation in rank order, one rounding to nearest even) |
    int32 MAX, for shards that straddle both region borders -- bit for bit against the torch path the gloo tests run
    (multiview.ViewParallelExchange._combine_compact), incl. half_end == sum_len (then it is r3dgs_reduce_shards)."""
    import torch
    from diff_gaussian_rasterization import _C
    world, shard = 8, 1000
    g = torch.Generator().manual_seed(11)
    for begin, sum_len, half_end in ((0, 400, 800), (2 * shard, 2 * shard + 17, 2 * shard + 18), (shard, 0, 5 * shard),
                                     (4 * shard, 4 * shard + 600, 4 * shard + 600), (6 * shard, 100, 200)):
        n_sum = min(max(sum_len - begin, 0), shard)
        n_half = min(max(half_end - begin, 0), shard)
        recv = torch.randn(world, shard, generator=g)
        if n_half > n_sum:
            halves = (torch.randn(world, 2 * (n_half - n_sum), generator=g) * 3).to(torch.bfloat16)
            recv[:, n_sum:n_half] = halves.view(torch.float32)
        ints = torch.randint(0, 500, (world, shard - n_half), generator=g, dtype=torch.int32)
        recv[:, n_half:] = ints.view(torch.float32)
        out = torch.empty(shard, device="cuda")
        _C.reduce_shards_mixed(recv.cuda().contiguous().view(-1), world, begin, sum_len, half_end, out)
        out = out.cpu()
        acc = recv[0, :n_sum].clone()
        for w in range(1, world):
            acc += recv[w, :n_sum]
        assert torch.equal(out[:n_sum], acc)
        if n_half > n_sum:
            h = recv[:, n_sum:n_half].contiguous().view(torch.bfloat16).view(world, -1).to(torch.float32)
            acc = h[0].clone()
            for w in range(1, world):
                acc += h[w]
            assert torch.equal(out[n_sum:n_half].contiguous().view(torch.bfloat16), acc.to(torch.bfloat16))
        assert torch.equal(out[n_half:].contiguous().view(torch.int32), ints.amax(0))
