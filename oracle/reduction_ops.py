"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's "reduction" operators that sit either side
of the rasterizer.  Nothing in the product imports this file.

Restated (fp32 arithmetic, same operation order as the scalar code; GLM 0.9.9 evaluation order for the
matrix products, GLM itself is an un-vendored submodule of the reference):
  min_pixel_size                 reduced_3dgs.cu:240-264 + reduced_3dgs/redundancy_score.cu:46-97
  sphere_ellipsoid_intersection  reduced_3dgs.cu:205-238 + redundancy_score.cu:121-159, :185-207
  min_redundancy                 reduced_3dgs.cu:268-287 + redundancy_score.cu:6-27
  kmeans                         reduced_3dgs.cu:290-340 + reduced_3dgs/kmeans.cu:5-53, :73-105
  knn_query_bruteforce           what simple_knn.cu:523-660 knn_indexQ computes (K nearest of a candidate subset for a query subset)
  knn_bruteforce                 what submodules/simple-knn/simple_knn.cu:143-191 (mean of 3) and :393-466
                                 (K nearest, excluding the query itself) compute; the reference's slot ORDER is an
                                 artefact of its box traversal, the neighbour SET and the distances are the contract.

parity unpinned: the reference ships no fixtures for these operators and its CUDA build cannot run here; nvcc's
FMA contraction of the same expressions is compiler-chosen, so the float outputs are compared to a tolerance and the
threshold decisions are excluded where they are numerically ambiguous (the `ambig` outputs below).
"""
import numpy as np

F = np.float32


def _mat4_vec4(m, v):
    """GLM mat4 * vec4 on the raw 16 floats (column c = m[4c:4c+4]): (col0*x + col1*y) + (col2*z + col3*w)."""
    m = m.astype(F)
    c0, c1, c2, c3 = m[0:4], m[4:8], m[8:12], m[12:16]
    x, y, z, w = (v[:, i:i + 1].astype(F) for i in range(4))
    return ((c0[None] * x + c1[None] * y) + (c2[None] * z + c3[None] * w)).astype(F)


def min_pixel_size(w2ndc, w2ndc_inv, means3D, image_height, image_width, want_ambig=False):
    """-> [P,1] fp32, 10000 where no camera sees the centre.  ambig[p]: some camera's frustum test is within 1e-5."""
    P = means3D.shape[0]
    out = np.full((P,), 10000.0, F)
    ambig = np.zeros((P,), bool)
    p4 = np.concatenate([means3D.astype(F), np.ones((P, 1), F)], 1)
    for c in range(w2ndc.shape[0]):
        m = w2ndc[c].reshape(16)
        mi = w2ndc_inv[c].reshape(16)
        hom = _mat4_vec4(m, p4)
        pw = F(1.0) / (hom[:, 3] + F(0.0000001))
        proj = hom[:, :3] * pw[:, None]
        depth = proj[:, 2]
        lo = np.array([-1, -1, 0], F)
        inside = np.all(proj <= F(1.0), 1) & np.all(proj >= lo[None], 1)
        ambig |= np.any(np.abs(proj - F(1.0)) < 1e-5, 1) | np.any(np.abs(proj - lo[None]) < 1e-5, 1)
        H, W = int(image_height[c]), int(image_width[c])
        end = np.zeros((P, 4), F)
        if W > H:
            end[:, 0] = F(2.0) / F(W)
        else:
            end[:, 1] = F(2.0) / F(H)
        end[:, 2] = depth
        end[:, 3] = 1
        start = np.zeros((P, 4), F)
        start[:, 2] = depth
        start[:, 3] = 1
        with np.errstate(all="ignore"):
            e = _mat4_vec4(mi, end)
            e3 = e[:, :3] * (F(1.0) / (e[:, 3] + F(0.0000001)))[:, None]
            s = _mat4_vec4(mi, start)
            s3 = s[:, :3] * (F(1.0) / (s[:, 3] + F(0.0000001)))[:, None]
            d = e3 - s3
            length = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F)
        # CUDA min(float, float): a NaN length leaves the old value (fminf semantics)
        upd = inside & ~np.isnan(length)
        out[upd] = np.minimum(out[upd], length[upd])
    if want_ambig:
        return out.reshape(P, 1), ambig
    return out.reshape(P, 1)


def rotation_columns(rot):
    """buildRotationMatrixCUDA (redundancy_score.cu:185-207): the three GLM columns, each [P,3]."""
    r, x, y, z = (rot[:, i].astype(F) for i in range(4))
    c0 = np.stack([F(1) - F(2) * (y * y + z * z), F(2) * (x * y + r * z), F(2) * (x * z - r * y)], 1)
    c1 = np.stack([F(2) * (x * y - r * z), F(1) - F(2) * (x * x + z * z), F(2) * (y * z + r * x)], 1)
    c2 = np.stack([F(2) * (x * z + r * y), F(2) * (y * z - r * x), F(1) - F(2) * (x * x + y * y)], 1)
    return c0.astype(F), c1.astype(F), c2.astype(F)


def sphere_ellipsoid_intersection(means3D, scales, rotations, nbr, radius, knn, ambig_rel=1e-5):
    """-> (count int32[P,1], mask bool[P,knn], ambig bool[P,knn]).  The rotation is the QUERY's own
    (`R[idx]`, redundancy_score.cu:147), the scales are the neighbour's."""
    P = means3D.shape[0]
    nbr = nbr.reshape(P, knn)
    c0, c1, c2 = rotation_columns(rotations)
    xyz = means3D.astype(F)
    diff = xyz[:, None, :] - xyz[nbr]                                  # [P,k,3]
    aug = scales.astype(F)[nbr] + radius.reshape(P, 1, 1).astype(F)    # [P,k,3]

    def vdot(col):                                                     # glm::dot(vec3, vec3): (x*x' + y*y') + z*z'
        return (col[:, None, 0] * diff[..., 0] + col[:, None, 1] * diff[..., 1]) + col[:, None, 2] * diff[..., 2]

    loc = np.stack([vdot(c0), vdot(c1), vdot(c2)], -1).astype(F)
    num = (loc * loc).astype(F)
    with np.errstate(all="ignore"):
        inv = (F(1.0) / (aug * aug).astype(F)).astype(F)
        val = ((num[..., 0] * inv[..., 0] + num[..., 1] * inv[..., 1]) + num[..., 2] * inv[..., 2]).astype(F)
    mask = val < F(1.0)
    ambig = np.abs(val - 1.0) < ambig_rel
    return mask.sum(1).astype(np.int32).reshape(P, 1), mask, ambig


def min_redundancy(redundancy, nbr, mask, knn):
    """-> int32[P,1], initial value P, scatter-min of the source's value onto each intersecting neighbour."""
    P = redundancy.shape[0]
    out = np.full((P,), P, np.int32)
    nbr = nbr.reshape(P, knn)
    src = np.broadcast_to(redundancy.reshape(P, 1).astype(np.int32), (P, knn))
    m = mask.reshape(P, knn)
    np.minimum.at(out, nbr[m], src[m])
    return out.reshape(P, 1)


def kmeans_assign(values, centers):
    """updateIdsCUDA (kmeans.cu:73-105): first centre with strictly smaller sqrt((c-v)^2)."""
    v = values.reshape(-1).astype(F)
    c = centers.reshape(-1).astype(F)
    ids = np.zeros(v.shape[0], np.int32)
    best = np.full(v.shape[0], np.inf, F)
    for i in range(c.shape[0]):
        with np.errstate(all="ignore"):
            d = np.sqrt(((c[i] - v) * (c[i] - v)).astype(F)).astype(F)
        upd = d < best
        best[upd] = d[upd]
        ids[upd] = i
    return ids


def kmeans(values, centers, tol, max_iterations):
    """-> (ids int32[n,1], centers fp32[n_centers], iterations run).  Sums in float64 then rounded: the reference
    adds fp32 partials in a nondeterministic atomic order, so its centres are only defined to ~1e-6 relative."""
    v = values.reshape(-1).astype(F)
    new = centers.reshape(-1).astype(F).copy()
    n_c = new.shape[0]
    it = 0
    for it in range(1, max_iterations + 1):
        ids = kmeans_assign(v, new)
        old = new.copy()
        sums = np.bincount(ids, weights=v.astype(np.float64), minlength=n_c).astype(F)
        sizes = np.bincount(ids, minlength=n_c)
        with np.errstate(all="ignore"):
            new = (sums / sizes.astype(F)).astype(F)
        new[np.isnan(new)] = 0
        if float(np.abs(old - new).astype(F).sum(dtype=F)) < tol:
            break
    ids = kmeans_assign(v, new)
    return ids.reshape(-1, 1), new, it


def knn_bruteforce(points, K, block=2048):
    """-> (d2 fp32[P,K] ascending, idx int32[P,K]); the query itself (same index) is excluded, duplicates are not.
    Slots that cannot be filled (P-1 < K) hold FLT_MAX / -1 like the reference's initial values."""
    p = points.astype(F)
    P = p.shape[0]
    d2o = np.full((P, K), np.finfo(F).max, F)
    ido = np.full((P, K), -1, np.int32)
    kk = min(K, P - 1)
    if kk <= 0:
        return d2o, ido
    for s in range(0, P, block):
        q = p[s:s + block]
        dx = p[None, :, 0] - q[:, None, 0]
        dy = p[None, :, 1] - q[:, None, 1]
        dz = p[None, :, 2] - q[:, None, 2]
        d2 = ((dx * dx + dy * dy) + dz * dz).astype(F)
        d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf
        part = np.argpartition(d2, kk - 1, axis=1)[:, :kk]
        # argpartition picks arbitrarily among equal distances at the K-th place: redo those rows by (distance, index)
        kth = np.take_along_axis(d2, part, 1).max(1)
        for r in np.nonzero((d2 <= kth[:, None]).sum(1) > kk)[0]:
            cand = np.nonzero(d2[r] <= kth[r])[0]
            part[r] = cand[np.lexsort((cand, d2[r, cand]))[:kk]]
        pd = np.take_along_axis(d2, part, 1)
        order = np.lexsort((part, pd), axis=1)
        d2o[s:s + block, :kk] = np.take_along_axis(pd, order, 1)
        ido[s:s + block, :kk] = np.take_along_axis(part, order, 1)
    return d2o, ido


def knn_query_bruteforce(points, q_indices, n_indices, K):
    """distIndexQ (simple-knn/spatial.cu:43-58, simple_knn.cu:523-660): for query q = points[q_indices[q]], the K nearest
    among the SET of points whose index is in n_indices (the reference marks them in a bool mask, so duplicates count
    once), the query's own index excluded.  -> (d2 fp32[Q,K], idx int32[Q,K]) ascending by (distance, index); unfilled
    slots FLT_MAX / -1; indices outside [0, P) are ignored."""
    p = points.astype(F)
    P = p.shape[0]
    Q = len(q_indices)
    d2o = np.full((Q, K), np.finfo(F).max, F)
    ido = np.full((Q, K), -1, np.int32)
    cand = np.unique(np.asarray([n for n in n_indices if 0 <= n < P], np.int64))
    for r, qi in enumerate(q_indices):
        if not (0 <= qi < P) or cand.size == 0:
            continue
        c = cand[cand != qi]
        dx, dy, dz = p[c, 0] - p[qi, 0], p[c, 1] - p[qi, 1], p[c, 2] - p[qi, 2]
        d2 = ((dx * dx + dy * dy) + dz * dz).astype(F)
        order = np.lexsort((c, d2))[:K]
        d2o[r, :order.size] = d2[order]
        ido[r, :order.size] = c[order]
    return d2o, ido


def knn_mean_dist3(points):
    """distCUDA2 (simple_knn.cu:143-191): (best0 + best1 + best2) / 3 of the squared distances."""
    d2, _ = knn_bruteforce(points, 3)
    with np.errstate(all="ignore"):
        return (((d2[:, 0] + d2[:, 1]) + d2[:, 2]) / F(3.0)).astype(F)
