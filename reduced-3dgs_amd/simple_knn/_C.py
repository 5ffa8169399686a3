"""`simple_knn._C` -- same symbols as the reference's pybind module (submodules/simple-knn/ext.cpp:15-19),
marshalled onto r3dgs_knn (include/r3dgs_reduction.h).  No CPU path: tensors must live on a GPU.

  distCUDA2(points)        -> mean squared distance to the 3 nearest other points, fp32[P]   (spatial.cu:15-27)
  distIndex2(points, K)    -> (dists fp32[P*K], indices int32[P*K])                           (spatial.cu:29-43)
     The reference leaves each point's K slots in the order its box traversal filled them; here every row is
     ascending by (distance, index).  The neighbour set and the distances are the same.
  distIndexQ(points, q_indices, n_indices, K) -> (dists fp32[Q*K], indices int32[Q*K]): K nearest among the points listed in
     n_indices for the query points listed in q_indices (spatial.cu:43-58); rows ascending by (distance, index), unfilled
     slots FLT_MAX / -1.  Exact tiled scan, O(Q * N): the reference tree never calls it.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _C as _r

_lib = _r._lib
_lib.r3dgs_knn_max_k.restype = C.c_int
_lib.r3dgs_knn_workspace_bytes.restype = C.c_size_t
_lib.r3dgs_knn_workspace_bytes.argtypes = [C.c_int]
_lib.r3dgs_knn.restype = C.c_int
_lib.r3dgs_knn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6


def _run(points, K, want_mean):
    if not points.is_cuda:
        raise RuntimeError("simple_knn: points must be on a GPU (no CPU path)")
    if points.dim() != 2 or points.size(1) != 3 or points.dtype != torch.float32:
        raise RuntimeError("simple_knn: points must be float32 of shape (P, 3)")
    dev = points.device
    pts = points.contiguous()
    P = int(pts.size(0))
    if want_mean:
        mean = torch.zeros((P,), dtype=torch.float32, device=dev)
        dists = indices = None
    else:
        K = int(K)
        if K < 1 or K > _lib.r3dgs_knn_max_k():
            raise RuntimeError(f"simple_knn: K must be in [1, {_lib.r3dgs_knn_max_k()}]")
        mean = None
        dists = torch.empty((P * K,), dtype=torch.float32, device=dev)
        indices = torch.empty((P * K,), dtype=torch.int32, device=dev)
    if P:
        with torch.cuda.device(dev):
            nbytes = _lib.r3dgs_knn_workspace_bytes(P)
            if nbytes == 0:
                raise RuntimeError(f"simple_knn: {_lib.r3dgs_last_error().decode()}")
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            _r._check(_lib.r3dgs_knn(P, 3 if want_mean else K, pts.data_ptr(), _r._ptr(dists), _r._ptr(indices),
                                     _r._ptr(mean), ws.data_ptr(), _r._stream()), "simple_knn")
    return mean if want_mean else (dists, indices)


def distCUDA2(points):
    return _run(points, 3, True)


def distIndex2(points, K):
    d, i = _run(points, K, False)
    return [d, i]


_lib.r3dgs_knn_query_workspace_bytes.restype = C.c_size_t
_lib.r3dgs_knn_query_workspace_bytes.argtypes = [C.c_int]
_lib.r3dgs_knn_query.restype = C.c_int
_lib.r3dgs_knn_query.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5


def distIndexQ(points, q_indices, n_indices, K):
    if not points.is_cuda:
        raise RuntimeError("simple_knn: points must be on a GPU (no CPU path)")
    if points.dim() != 2 or points.size(1) != 3 or points.dtype != torch.float32:
        raise RuntimeError("simple_knn: points must be float32 of shape (P, 3)")
    dev = points.device
    for name, t in (("q_indices", q_indices), ("n_indices", n_indices)):
        if t.device != dev or t.dtype != torch.int32:
            raise RuntimeError(f"simple_knn: {name} must be int32 on {dev}")
    K = int(K)
    if K < 1 or K > 4096:
        raise RuntimeError("simple_knn: K must be in [1, 4096]")
    pts, qi, ni = points.contiguous(), q_indices.contiguous().reshape(-1), n_indices.contiguous().reshape(-1)
    P, Q, N = int(pts.size(0)), int(qi.numel()), int(ni.numel())
    dists = torch.empty((Q * K,), dtype=torch.float32, device=dev)
    indices = torch.empty((Q * K,), dtype=torch.int32, device=dev)
    if Q:
        with torch.cuda.device(dev):
            ws = torch.empty((_lib.r3dgs_knn_query_workspace_bytes(P),), dtype=torch.uint8, device=dev)
            _r._check(_lib.r3dgs_knn_query(P, K, _r._ptr(pts), Q, qi.data_ptr(), N, _r._ptr(ni), dists.data_ptr(),
                                           indices.data_ptr(), ws.data_ptr(), _r._stream()), "simple_knn")
    return [dists, indices]
