"""`diff_gaussian_rasterization._C` -- operator set of the MI355X-native rasterizer.

Same symbols, positional signatures, return tuples and error behaviour as the reference's pybind
module (submodules/diff-gaussian-rasterization/ext.cpp:16-25, rasterize_points.h:18-92), implemented as
thin marshalling over the C ABI of libr3dgs_hip.so (include/r3dgs_rasterizer.h).  PyTorch is plumbing
only here: device memory (outputs + the three opaque state blobs), the current HIP stream, contiguity.

There is NO fallback: if the HIP library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import threading
import weakref
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "libr3dgs_hip.so")
if os.environ.get("R3DGS_LIB"):   # A/B builds (tools/build_variant.py): a path, or the tag of libr3dgs_hip_<tag>.so
    _v = os.environ["R3DGS_LIB"]
    _LIB_PATH = _v if os.sep in _v else os.path.join(os.path.dirname(_HERE), f"libr3dgs_hip_{_v}.so")
if not os.path.exists(_LIB_PATH):
    raise ImportError(f"{_LIB_PATH} not found: build it with `python reduced-3dgs_amd/build.py` "
                      "(the rasterizer has no CPU or PyTorch fallback)")
_lib = C.CDLL(_LIB_PATH)

_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

_lib.r3dgs_version.restype = C.c_char_p
_lib.r3dgs_last_error.restype = C.c_char_p
_lib.r3dgs_geometry_bytes.restype = C.c_size_t
_lib.r3dgs_geometry_bytes.argtypes = [_i]
_lib.r3dgs_geometry_bytes_lean.restype = C.c_size_t
_lib.r3dgs_geometry_bytes_lean.argtypes = [_i]
_lib.r3dgs_binning_bytes.restype = C.c_size_t
_lib.r3dgs_binning_bytes.argtypes = [_i, _i, _i, _i]
_lib.r3dgs_image_bytes.restype = C.c_size_t
_lib.r3dgs_image_bytes.argtypes = [_i, _i]
_lib.r3dgs_binning_capacity.restype = _i
_lib.r3dgs_binning_capacity.argtypes = [_i, _i, _i, C.c_size_t]
_lib.r3dgs_mark_visible.restype = _i
_lib.r3dgs_mark_visible.argtypes = [_i, _vp, _vp, _vp, _vp, _vp]
_FWD_TAIL = [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _i,
             _i, _vp]
_lib.r3dgs_forward.restype = _i
_lib.r3dgs_forward.argtypes = [_ALLOC, _vp, _ALLOC, _vp, _ALLOC, _vp, _i, _vp, _i] + _FWD_TAIL
_lib.r3dgs_inference_forward.restype = _i
_lib.r3dgs_inference_forward.argtypes = [_ALLOC, _vp, _ALLOC, _vp, _ALLOC, _vp, _i, _vp, _i, _vp, _vp, _vp] + _FWD_TAIL
_lib.r3dgs_reserve_hint.restype = _i
_lib.r3dgs_reserve_hint.argtypes = [_i, _i, _i]
_lib.r3dgs_reserve_hint_view.restype = _i
_lib.r3dgs_reserve_hint_view.argtypes = [_i, _i, _i, _vp]
_lib.r3dgs_forward_reserved.restype = C.c_longlong
_lib.r3dgs_forward_reserved.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _i] + _FWD_TAIL
_lib.r3dgs_inference_forward_reserved.restype = C.c_longlong
_lib.r3dgs_inference_forward_reserved.argtypes = [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp] + _FWD_TAIL
_lib.r3dgs_pass_query.restype = _i
_lib.r3dgs_pass_query.argtypes = [C.c_longlong, _i, _vp, _vp, _vp, _vp]
_lib.r3dgs_pass_pairs.restype = _i
_lib.r3dgs_pass_pairs.argtypes = [C.c_longlong, _i]
_lib.r3dgs_forward_pairs.restype = _i
_lib.r3dgs_set_tight_rects.restype = _i
_lib.r3dgs_set_tight_rects.argtypes = [_i]
_lib.r3dgs_forward_hint.restype = None
_lib.r3dgs_forward_hint.argtypes = [_i]
_lib.r3dgs_set_sh_cache.restype = _i
_lib.r3dgs_set_sh_cache.argtypes = [_i]
_lib.r3dgs_reserve_forget_view.restype = None
_lib.r3dgs_reserve_forget_view.argtypes = [C.c_void_p]
_lib.r3dgs_set_f64_chain.restype = _i
_lib.r3dgs_set_f64_chain.argtypes = [_i]
_lib.r3dgs_set_tile_order.restype = _i
_lib.r3dgs_set_tile_order.argtypes = [_i]
_lib.r3dgs_export_tile_order.restype = _i
_lib.r3dgs_export_tile_order.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
if hasattr(_lib, "r3dgs_bwd_units_cap"):   # (absent from an older A/B build loaded through R3DGS_LIB)
    _lib.r3dgs_bwd_units_cap.restype = _i
    _lib.r3dgs_bwd_units_cap.argtypes = [_i, _i, _i]
    _lib.r3dgs_set_bwd_segments.restype = _i
    _lib.r3dgs_set_bwd_segments.argtypes = [_i]
_lib.r3dgs_export_rects.restype = _i
_lib.r3dgs_export_rects.argtypes = [_i, _vp, _vp, _vp]
_lib.r3dgs_reserve_overflow_events.restype = C.c_longlong
_lib.r3dgs_reserve_overflow_events.argtypes = [_vp, _vp]
_lib.r3dgs_backward.restype = _i
_lib.r3dgs_backward.argtypes = ([_i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                 _vp, _vp, _vp, _vp] + [_vp] * 10 + [_f, _i, _vp])
_lib.r3dgs_export_binning.restype = _i
_lib.r3dgs_export_binning.argtypes = [_i, _i, _i, _i, _i] + [_vp] * 10

_lib.r3dgs_colour_variance_accumulate.restype = _i
_lib.r3dgs_colour_variance_accumulate.argtypes = [_i, _vp, _i, _i] + [_vp] * 12
_lib.r3dgs_min_pixel_size.restype = _i
_lib.r3dgs_min_pixel_size.argtypes = [_i, _i] + [_vp] * 7
_lib.r3dgs_sphere_ellipsoid_intersection.restype = _i
_lib.r3dgs_sphere_ellipsoid_intersection.argtypes = [_i, _i] + [_vp] * 8
_lib.r3dgs_min_redundancy.restype = _i
_lib.r3dgs_min_redundancy.argtypes = [_i, _i] + [_vp] * 5
_lib.r3dgs_kmeans_workspace_bytes.restype = C.c_size_t
_lib.r3dgs_kmeans_workspace_bytes.argtypes = [_i, _i]
_lib.r3dgs_kmeans.restype = _i
_lib.r3dgs_kmeans.argtypes = [_i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]
_lib.r3dgs_pack_view_stats.restype = _i
_lib.r3dgs_pack_view_stats.argtypes = [_i] + [_vp] * 6
_lib.r3dgs_reduce_shards.restype = _i
_lib.r3dgs_reduce_shards.argtypes = [_i, C.c_longlong, C.c_longlong, C.c_longlong, _vp, _vp, _vp]
if hasattr(_lib, "r3dgs_reduce_shards_mixed"):   # (absent from an older A/B build loaded through R3DGS_LIB)
    _lib.r3dgs_reduce_shards_mixed.restype = _i
    _lib.r3dgs_reduce_shards_mixed.argtypes = [_i, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _vp, _vp, _vp]
_lib.r3dgs_profile_enable.argtypes = [_i]
_lib.r3dgs_profile_stage_name.restype = C.c_char_p
_lib.r3dgs_profile_stage_name.argtypes = [_i]
_lib.r3dgs_profile_read.argtypes = [_vp, _vp]

LIBRARY_PATH = _LIB_PATH

# ---- compiled torch binding of the two hot calls (csrc_torch/r3dgs_torch.cpp; the reference's layer is a torch C++
# extension too, DGR/ext.cpp:16-25).  It is handed the entry points of the library loaded above, so both bindings drive
# ONE library instance.  R3DGS_BINDING = auto (default: compiled if it was built, else ctypes) | torch (required) | ctypes.
_EXT_FUNCS = ("r3dgs_last_error", "r3dgs_version", "r3dgs_geometry_bytes", "r3dgs_geometry_bytes_lean", "r3dgs_binning_bytes",
              "r3dgs_image_bytes", "r3dgs_forward_hint", "r3dgs_reserve_hint_view", "r3dgs_forward_reserved",
              "r3dgs_pass_query", "r3dgs_backward", "r3dgs_mark_visible")
_ext = None
_ext_loaded = None
_binding_request = os.environ.get("R3DGS_BINDING", "auto")
if _binding_request not in ("auto", "torch", "ctypes"):
    raise ImportError(f"R3DGS_BINDING={_binding_request!r}: expected auto, torch or ctypes")
if _binding_request != "ctypes":
    try:
        from . import _r3dgs_torch as _ext_loaded
        _ext_loaded.bind({n: C.cast(getattr(_lib, n), C.c_void_p).value for n in _EXT_FUNCS})
        _ext = _ext_loaded
    except ImportError:
        if _binding_request == "torch":
            raise
        _ext_loaded = None


def binding():
    """'torch': rasterize_gaussians / rasterize_gaussians_backward / mark_visible go through the compiled torch extension
    (diff_gaussian_rasterization/_r3dgs_torch.so); 'ctypes': through this module's ctypes marshalling.  Same library, same
    kernels, same results bit for bit (tests/test_gpu_parity.py runs both)."""
    return "torch" if _ext is not None else "ctypes"


def set_binding(name):
    """'torch' | 'ctypes' -> the previous setting.  'torch' needs the extension to have been built (build.py)."""
    global _ext
    before = binding()
    if name == "torch":
        if _ext_loaded is None:
            raise RuntimeError("the compiled torch binding is not built (reduced-3dgs_amd/build.py) or R3DGS_BINDING=ctypes")
        _ext = _ext_loaded
    elif name == "ctypes":
        _ext = None
    else:
        raise ValueError(name)
    return before


_NO_TENSOR = torch.Tensor([])


def _t(x):
    return _NO_TENSOR if x is None else x



def profile_enable(on, only=None):
    """Per-stage HIP-event timing inside the library (include/r3dgs_rasterizer.h r3dgs_profile_*).
    only: iterable of stage names to time alone (every event record is a packet on the stream)."""
    if on and only:
        n = _lib.r3dgs_profile_stage_count()
        names = [_lib.r3dgs_profile_stage_name(k).decode() for k in range(n)]
        code = 0
        for name in only:
            code |= 1 << (names.index(name) + 1)
        _lib.r3dgs_profile_enable(code)
    else:
        _lib.r3dgs_profile_enable(int(bool(on)))


def profile_read():
    """-> {stage_name: (total_ms, launches)} since the last read; waits for the recorded events."""
    n = _lib.r3dgs_profile_stage_count()
    ms = (C.c_double * n)()
    cnt = (C.c_int * n)()
    _check(_lib.r3dgs_profile_read(ms, cnt), "profile_read")
    return {_lib.r3dgs_profile_stage_name(k).decode(): (ms[k], cnt[k]) for k in range(n)}


def version():
    return _lib.r3dgs_version().decode()


def _check(status, what):
    if status < 0:
        raise RuntimeError(f"{what}: {_lib.r3dgs_last_error().decode()}")
    return status


def _ptr(t):
    """Device pointer of a tensor, or NULL for an absent optional input (empty tensor, as the reference's
    wrapper passes `torch.Tensor([])` for missing arguments, diff_gaussian_rasterization/__init__.py:209-218)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _dev_f32(t, dev):
    if t is None or t.numel() == 0:
        return None
    if t.device != dev:
        raise RuntimeError(f"expected a tensor on {dev}, got {t.device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def _dev_i32(t, dev):
    if t is None or t.numel() == 0:
        return None
    if t.device != dev:
        raise RuntimeError(f"expected a tensor on {dev}, got {t.device}")
    if t.dtype != torch.int32:
        raise RuntimeError(f"expected int32, got {t.dtype}")
    return t.contiguous()


class _Blob:
    """One caller-owned, resizable byte buffer (resizeFunctional of rasterize_points.cu:33-41)."""

    def __init__(self, dev):
        self.dev = dev
        self.tensor = torch.empty(0, dtype=torch.uint8, device=dev)
        self.error = None

        def alloc(nbytes, _user):
            try:
                self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
                return self.tensor.data_ptr()
            except Exception as e:  # an exception cannot cross the C frame: keep it, hand back NULL
                self.error = e
                return None

        self.cb = _ALLOC(alloc)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs ~10 us)."""

    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


class NumRendered:
    """The `num_rendered` slot of the forward's return tuple (rasterize_points.cu:221), with the reference's meaning:
    the sum over the visible Gaussians of the tile count of their 3-sigma squares.  The asynchronous forward does not
    wait for the number: it lives on the device, and this object fetches it the first time someone looks (int(),
    index, comparison, print) -- the training loop never does, it only hands the object back to
    rasterize_gaussians_backward.  `.capacity` is the pair capacity the binning buffer was sized with, `.pairs` the
    (tile, Gaussian) pairs the pass binned (<= num_rendered: opacity-aware rects leave out tiles a Gaussian cannot
    reach; equal with set_tight_rects(False))."""

    __slots__ = ("ticket", "capacity", "_value", "_flags", "_pairs")

    def __init__(self, ticket, capacity, value=None, pairs=None):
        self.ticket, self.capacity, self._value, self._flags, self._pairs = ticket, capacity, value, 0, pairs

    def _resolve(self, wait=True):
        if self._value is None:
            r, v, cap, fl = _i(), _i(), _i(), _i()
            st = _check(_lib.r3dgs_pass_query(self.ticket, int(wait), C.byref(r), C.byref(v), C.byref(cap),
                                              C.byref(fl)), "num_rendered")
            if st == 1:
                self._value, self._flags = r.value, fl.value
        return self._value

    @property
    def pairs(self):
        if self._pairs is None:
            self._pairs = _check(_lib.r3dgs_pass_pairs(self.ticket, 1), "pairs")
        return self._pairs

    @property
    def truncated(self):
        """True if num_rendered exceeded the reservation of this pass (the farthest pairs were dropped)."""
        self._resolve()
        return bool(self._flags & 1)

    def ready(self):
        return self._resolve(wait=False) is not None

    def __int__(self):
        return self._resolve()

    __index__ = __int__

    def __float__(self):
        return float(self._resolve())

    def __eq__(self, other):
        return int(self) == other

    def __lt__(self, other):
        return int(self) < other

    def __le__(self, other):
        return int(self) <= other

    def __gt__(self, other):
        return int(self) > other

    def __ge__(self, other):
        return int(self) >= other

    def __hash__(self):
        return hash(int(self))

    def __bool__(self):
        return int(self) != 0

    def __repr__(self):
        return str(int(self))


_size_cache = {}


def _blob_bytes(kind, *key):
    k = (kind,) + key
    v = _size_cache.get(k)
    if v is None:
        if kind == "geom":
            v = _lib.r3dgs_geometry_bytes(*key)
        elif kind == "geom_lean":
            v = _lib.r3dgs_geometry_bytes_lean(*key)
        elif kind == "bin":
            v = _lib.r3dgs_binning_bytes(*key)
        else:
            v = _lib.r3dgs_image_bytes(*key)
        if v == 0:
            raise RuntimeError(f"rasterize_gaussians: {_lib.r3dgs_last_error().decode()}")
        if len(_size_cache) > 4096:
            _size_cache.clear()
        _size_cache[k] = v
    return v


# ---- strict mode (the default) -------------------------------------------------------------------------------------
# The asynchronous forward runs on a pair RESERVATION.  The reference never drops a pair (it sizes its buffer from the
# exact count, rasterizer_impl.cu:441-450), so a pass whose pair count exceeded its reservation must not reach a
# consumer: in strict mode the forward looks at the pass's published numbers before it returns -- they appear ~40 us
# into the pass, long before it ends, so the GPU keeps working on the rest of the pass while the host goes on -- and a
# truncated pass is redone on the exact-size path (same stream, same output tensors: the redo overwrites them in
# stream order).  What the caller gets is therefore always the exact result; the reservation only decides how often a
# redo happens.  R3DGS_STRICT=0 / set_strict(False) selects the old fire-and-forget behaviour (a RuntimeWarning after
# the fact), for host-side A/B measurements only.
_strict = os.environ.get("R3DGS_STRICT", "1") != "0"
_stats = {"reserved_passes": 0, "exact_passes": 0, "redone_passes": 0}
_overflow_seen = 0
_calls = 0


def set_strict(on):
    """True (default): a pass that overflowed its pair reservation is detected before its outputs are returned and
    redone on the exact-size path.  False: nothing waits; such a pass drops its farthest pairs and only warns."""
    global _strict
    _strict = bool(on)


def is_strict():
    return _strict


def pass_stats():
    """{reserved_passes, exact_passes, redone_passes, overflow_events}: how the forwards of this process were issued."""
    d = dict(_stats)
    d["overflow_events"] = reserve_overflow_events()
    return d


def reserve_forget():
    """Forget the pair counts learnt so far (new scene; tests): the next pass of each image size is an exact-size one."""
    _lib.r3dgs_reserve_forget()


def reserve_overflow_events():
    """Passes whose pair count exceeded their reservation so far (strict mode redid them; the reservation grows)."""
    return int(_lib.r3dgs_reserve_overflow_events(None, None))


def _watch_overflow():
    global _overflow_seen, _calls
    _calls += 1
    if _calls & 15:
        return
    r, cap = _i(), _i()
    n = int(_lib.r3dgs_reserve_overflow_events(C.byref(r), C.byref(cap)))
    if n > _overflow_seen:
        _overflow_seen = n
        import warnings
        warnings.warn(f"diff_gaussian_rasterization (strict mode OFF): a pass needed {r.value} (tile, Gaussian) pairs but "
                      f"{cap.value} were reserved; its farthest pairs were dropped and the reservation has been raised",
                      RuntimeWarning)


_tracked_views = {}   # device address of a view matrix -> weakref of the tensor that owns it


def _track_view(vm):
    """The library remembers pair counts per camera under the device address of its view matrix.  When that tensor dies
    the address may be handed to another camera: tell the library to forget it.  Every scene/cameras.py Camera
    owns its matrix for the whole run, so this registers once per camera; a matrix that is a temporary (uploaded per
    iteration, a converted copy) is forgotten as soon as it is freed and its passes use the image size's largest recent
    count instead."""
    ptr = vm.data_ptr()
    ref = _tracked_views.get(ptr)
    if ref is not None and ref() is vm:
        return
    if len(_tracked_views) > 65536:
        _tracked_views.clear()
    _tracked_views[ptr] = weakref.ref(vm)
    weakref.finalize(vm, _forget_view, ptr, id(vm))


def _forget_view(ptr, ident):
    ref = _tracked_views.get(ptr)
    if ref is not None and ref() is not None:
        # a live tensor has registered this address since (a fresh wrapper over the same storage every iteration --
        # stacked[i], .view() -- dies while its successor is in use): the camera's learnt pair count stays
        return
    if ref is not None:
        del _tracked_views[ptr]
    try:
        _lib.r3dgs_reserve_forget_view(ptr)
    except Exception:   # interpreter shutdown
        pass


def _forward_common(ragged, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                    viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degrees, campos,
                    prefiltered, debug, counters=None, exact=False, _reserve=None, _strict_override=None):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:158-161
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("the MI355X rasterizer needs device tensors (no CPU path)")
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    u8 = dict(dtype=torch.uint8, device=dev)
    if P == 0:
        out_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.zeros((0,), dtype=torch.int32, device=dev)
        e = torch.empty(0, **u8)
        return NumRendered(0, 0, 0, 0), out_color, radii, e, e.clone(), e.clone()
    hint = getattr(_tls, "next_forward_trains", None)   # per thread, like the library's own r3dgs_forward_hint state
    trains = torch.is_grad_enabled() if hint is None else hint
    _tls.next_forward_trains = None
    if _ext is not None and ragged is None and counters is None and not exact and not debug and _reserve is None:
        # the hot call: compiled marshalling (csrc_torch/r3dgs_torch.cpp), same library entry points as below
        strict = _strict if _strict_override is None else bool(_strict_override)
        if viewmatrix is not None and viewmatrix.is_contiguous() and viewmatrix.dtype == torch.float32:
            _track_view(viewmatrix)
        ticket, reserve, rendered, flags, out_color, radii, geom, binning, img = _ext.forward_reserved(
            _t(background), means3D, _t(colors), _t(opacity), _t(scales), _t(rotations), float(scale_modifier),
            _t(cov3D_precomp), _t(viewmatrix), _t(projmatrix), float(tan_fovx), float(tan_fovy), H, W, _t(sh), _t(degrees),
            _t(campos), bool(prefiltered), bool(trains), strict)
        if ticket:
            _stats["reserved_passes"] += 1
            nr = NumRendered(ticket, reserve, rendered if rendered >= 0 else None)
            nr._flags = flags
            if not strict:
                _watch_overflow()
                return nr, out_color, radii, geom, binning, img
            if not nr.truncated:
                return nr, out_color, radii, geom, binning, img
            _stats["redone_passes"] += 1   # the reservation did not hold: redo on the exact-size path below
            del geom, binning, img
        exact = True   # (ticket == 0: nothing known about this view size yet)
    out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    bg = _dev_f32(background, dev)
    m3, col, op = _dev_f32(means3D, dev), _dev_f32(colors, dev), _dev_f32(opacity, dev)
    sc, rot, cov = _dev_f32(scales, dev), _dev_f32(rotations, dev), _dev_f32(cov3D_precomp, dev)
    vm, pm, cp = _dev_f32(viewmatrix, dev), _dev_f32(projmatrix, dev), _dev_f32(campos, dev)
    shc, deg = _dev_f32(sh, dev), _dev_i32(degrees, dev)
    touched = transm = None
    if counters is not None:
        touched, transm = counters
    _lib.r3dgs_forward_hint(int(trains))   # holds for the redo of an overflowed pass as well
    with _on_device(dev):
        tail = (_ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(op), _ptr(sc), float(scale_modifier), _ptr(rot),
                _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                _ptr(out_color), _ptr(touched), _ptr(transm), _ptr(radii), int(counters is not None),
                int(bool(debug)), _stream())
        if ragged is None:
            head = (P, _ptr(deg), int(shc.size(1)) if shc is not None else 0)
            fn_exact, fn_reserved = _lib.r3dgs_forward, _lib.r3dgs_forward_reserved
        else:
            coeffs, perband, cumsum = (_dev_i32(t, dev) for t in ragged)
            head = (P, _ptr(deg), int(perband.numel()) if perband is not None else 0, _ptr(coeffs), _ptr(perband),
                    _ptr(cumsum))
            fn_exact, fn_reserved = _lib.r3dgs_inference_forward, _lib.r3dgs_inference_forward_reserved
        # Asynchronous path: blobs sized up front from a pair reservation, one graph launch, no wait for the pass.  The
        # exact-size path (allocator callbacks, one wait in the middle) runs when nothing is known about this view size
        # yet, in debug mode, when asked for -- and to redo a pass that overflowed its reservation (strict mode).
        if vm is not None:
            _track_view(vm)
        reserve = 0 if (exact or debug) else _lib.r3dgs_reserve_hint_view(P, W, H, _ptr(vm))
        if _reserve is not None:   # tests: a chosen reservation
            reserve = int(_reserve)
        strict = _strict if _strict_override is None else bool(_strict_override)
        if reserve > 0:
            # no SH direction derivatives will be left (render-only / ragged SH / precomputed colours): the lean blob
            lean = not trains or ragged is not None or shc is None or col is not None
            geom = torch.empty(_blob_bytes("geom_lean" if lean else "geom", P), **u8)
            binning = torch.empty(_blob_bytes("bin", P, W, H, reserve), **u8)
            img = torch.empty(_blob_bytes("img", W, H), **u8)
            ticket = fn_reserved(geom.data_ptr(), binning.data_ptr(), img.data_ptr(), reserve, *head, *tail)
            if ticket < 0:
                _check(-1, "rasterize_gaussians")
            _stats["reserved_passes"] += 1
            nr = NumRendered(ticket, reserve)
            if not strict:
                _watch_overflow()
                return nr, out_color, radii, geom, binning, img
            if not nr.truncated:   # waits for the pass's header (not for the pass)
                return nr, out_color, radii, geom, binning, img
            # The reservation did not hold: redo on the exact-size path.  Counter mode accumulates into its outputs.
            _stats["redone_passes"] += 1
            if counters is not None:
                touched.zero_()
                transm.zero_()
            del geom, binning, img
        geom, binning, img = _Blob(dev), _Blob(dev), _Blob(dev)
        rendered = fn_exact(geom.cb, None, binning.cb, None, img.cb, None, *head, *tail)
    for blob in (geom, binning, img):
        if blob.error is not None:   # e.g. torch OOM inside the allocator callback: surface the original exception
            raise blob.error
    _check(rendered, "rasterize_gaussians")
    _stats["exact_passes"] += 1
    pairs = int(_lib.r3dgs_forward_pairs())   # pairs actually binned (<= num_rendered, what the blob was carved for)
    return NumRendered(0, max(int(rendered), 1), rendered, pairs), out_color, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degrees, campos,
                        prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:136-222) ->
    (num_rendered, out_color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer).
    num_rendered is a NumRendered: an int-like that is only fetched from the device when looked at."""
    return _forward_common(None, background, means3D, colors, opacity, scales, rotations, scale_modifier,
                           cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                           degrees, campos, prefiltered, debug)


def rasterize_gaussians_variableSH_bands(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                         cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                         image_width, sh, perBandPrimitiveCount, cumSumPrimitiveCount, coeffsNum,
                                         degrees, campos, prefiltered, debug):
    """RasterizeGaussiansVariableSHBandsCUDA (rasterize_points.cu:43-134): inference-only forward over the
    ragged, degree-sorted SH buffer."""
    return _forward_common((coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount), background, means3D, colors,
                           opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                           tan_fovx, tan_fovy, image_height, image_width, sh, degrees, campos, prefiltered, debug)


_grad_arena = None


def set_gradient_arena(fn):
    """Extension for multi-GPU training (not in the reference): `fn(name, shape) -> float32 CUDA tensor or None`
    supplies the storage of the backward's outputs (names: means2D, colors, opacity, means3D, cov3D, sh, scales,
    rotations).  A view-parallel trainer hands out views of its flat all-reduce buffer, so the gradients are born
    inside it and no packing copy is needed (multiview.ViewParallelExchange.arena).  None restores torch.empty."""
    global _grad_arena
    _grad_arena = fn


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                 degrees, campos, geomBuffer, R, binningBuffer, imageBuffer, lambda_sh_sparsity,
                                 debug, _want_conic=False):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:224-305) -> (dL_dmeans2D[P,3], dL_dcolors[P,3],
    dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])."""
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    opts = dict(dtype=torch.float32, device=dev)
    if P == 0:
        z = lambda *s: torch.zeros(s, **opts)
        return z(0, 3), z(0, 3), z(0, 1), z(0, 3), z(0, 6), z(0, M, 3), z(0, 3), z(0, 4)
    if _ext is not None and _grad_arena is None:
        if isinstance(R, NumRendered):
            cap = R.capacity
        else:
            cap = _lib.r3dgs_binning_capacity(P, W, H, int(binningBuffer.numel())) if binningBuffer.numel() else 0
        return tuple(_ext.backward(_t(background), means3D, _t(radii), _t(colors), _t(scales), _t(rotations),
                                   float(scale_modifier), _t(cov3D_precomp), _t(viewmatrix), _t(projmatrix), float(tan_fovx),
                                   float(tan_fovy), dL_dout_color, _t(sh), _t(degrees), _t(campos), geomBuffer, int(cap),
                                   binningBuffer, imageBuffer, float(lambda_sh_sparsity), bool(debug), bool(_want_conic)))

    def e(name, *s):  # every element is written by the library
        if _grad_arena is not None:
            t = _grad_arena(name, s)
            if t is not None:
                if tuple(t.shape) != s or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                    raise RuntimeError(f"gradient arena returned an unusable tensor for {name}")
                return t
        return torch.empty(s, **opts)
    dL_dmeans3D, dL_dmeans2D = e("means3D", P, 3), e("means2D", P, 3)
    dL_dcolors, dL_dopacity = e("colors", P, 3), e("opacity", P, 1)
    dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = e("cov3D", P, 6), e("sh", P, M, 3), e("scales", P, 3), e("rotations", P, 4)
    dL_dconic = torch.empty((P, 2, 2), **opts) if _want_conic else None
    bg, m3 = _dev_f32(background, dev), _dev_f32(means3D, dev)
    col, sc, rot, cov = (_dev_f32(t, dev) for t in (colors, scales, rotations, cov3D_precomp))
    vm, pm, cp = _dev_f32(viewmatrix, dev), _dev_f32(projmatrix, dev), _dev_f32(campos, dev)
    g, shc, deg, rad = _dev_f32(dL_dout_color, dev), _dev_f32(sh, dev), _dev_i32(degrees, dev), _dev_i32(radii, dev)
    if isinstance(R, NumRendered):
        cap = R.capacity
    else:   # a plain int from an older caller: the capacity is what the binning buffer was sized with
        cap = _lib.r3dgs_binning_capacity(P, W, H, int(binningBuffer.numel())) if binningBuffer.numel() else 0
    with _on_device(dev):
        st = _lib.r3dgs_backward(P, _ptr(deg), M, int(cap), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(sc),
                                 float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cp),
                                 float(tan_fovx), float(tan_fovy), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer),
                                 _ptr(imageBuffer), _ptr(g), _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity),
                                 _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales),
                                 _ptr(dL_drotations), float(lambda_sh_sparsity), int(bool(debug)), _stream())
    _check(st, "rasterize_gaussians_backward")
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    return out + (dL_dconic,) if _want_conic else out


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:307-326): bool[P], view-space z > 0.2."""
    if _ext is not None:
        return _ext.mark_visible(means3D, _t(viewmatrix), _t(projmatrix))
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P:
        m3, vm, pm = _dev_f32(means3D, dev), _dev_f32(viewmatrix, dev), _dev_f32(projmatrix, dev)
        with _on_device(dev):
            _check(_lib.r3dgs_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), _ptr(present), _stream()), "mark_visible")
    return present


def export_binning(P, R, H, W, geomBuffer, binningBuffer, imageBuffer):
    """Debug accessor (not in the reference): sorted 64-bit keys in the reference's format, point list, tile
    ranges, n_contrib, final T and tiles_touched of a finished forward -- for bit-exact integer parity tests."""
    dev = geomBuffer.device
    gx, gy = (W + 15) // 16, (H + 15) // 16
    if isinstance(R, NumRendered):
        cap, R = R.capacity, R.pairs
    else:
        cap = _lib.r3dgs_binning_capacity(P, W, H, int(binningBuffer.numel())) if binningBuffer.numel() else 0
    R = min(int(R), cap)
    keys = torch.empty((R,), dtype=torch.int64, device=dev)
    plist = torch.empty((R,), dtype=torch.int32, device=dev)
    ranges = torch.empty((gx * gy, 2), dtype=torch.int32, device=dev)
    n_contrib = torch.empty((H * W,), dtype=torch.int32, device=dev)
    final_T = torch.empty((H * W,), dtype=torch.float32, device=dev)
    tiles = torch.empty((P,), dtype=torch.int32, device=dev)
    with _on_device(dev):
        _check(_lib.r3dgs_export_binning(P, cap, R, W, H, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                                         _ptr(keys), _ptr(plist), _ptr(ranges), _ptr(n_contrib), _ptr(final_T),
                                         _ptr(tiles), _stream()), "export_binning")
    return dict(keys=keys, point_list=plist, ranges=ranges, n_contrib=n_contrib, final_T=final_T, tiles_touched=tiles)


def export_rects(P, geomBuffer):
    """Debug accessor (not in the reference): the tile rect (x0, y0, x1, y1; exclusive maxima) every Gaussian was binned
    into, int32[P,4] (rows of culled Gaussians are undefined)."""
    out = torch.empty((P, 4), dtype=torch.int16, device=geomBuffer.device)
    with _on_device(geomBuffer.device):
        _check(_lib.r3dgs_export_rects(P, _ptr(geomBuffer), _ptr(out), _stream()), "export_rects")
    return out.to(torch.int32) & 0xFFFF


def set_tight_rects(on):
    """True (default): a Gaussian is binned into the tiles of the reference's 3-sigma square that it can actually reach
    with alpha >= 1/255 (bounding box of that ellipse); False: into the whole square, i.e. lists identical to the
    reference's.  Image, radii, num_rendered and gradients are the same either way.  Returns the previous setting."""
    return bool(_lib.r3dgs_set_tight_rects(int(bool(on))))


def tight_rects():
    return bool(_lib.r3dgs_set_tight_rects(-1))


_tls = threading.local()   # .next_forward_trains: None / absent = nobody said -- a forward under torch.no_grad() is a rendering


def hint_next_forward(will_backward):
    """The autograd wrapper says whether any input of the forward it is about to issue needs a gradient.  A forward that
    no backward will follow (rendering under no_grad) skips what it would only do for the backward's sake (the SH direction
    derivatives, 36 B per visible Gaussian).  One-shot: the call after the next forward trains again."""
    _tls.next_forward_trains = bool(will_backward)


def set_sh_cache(on):
    """True (default): a backward without a sparsity term uses the SH direction derivatives the forward left instead of
    reading the SH tensor again; False: it reads the rows.  Same gradients, bit for bit.  Returns the previous setting."""
    return bool(_lib.r3dgs_set_sh_cache(int(bool(on))))


def sh_cache():
    return bool(_lib.r3dgs_set_sh_cache(-1))


def set_f64_chain(on):
    """True (default): the per-Gaussian backward evaluates the covariance chain (backward.cu:228-306, 311-374) in double and
    rounds once; False: in fp32, as the reference does.  Returns the previous setting."""
    return bool(_lib.r3dgs_set_f64_chain(int(bool(on))))


def f64_chain():
    return bool(_lib.r3dgs_set_f64_chain(-1))


def set_tile_order(on):
    """True (default): the backward blend starts its tiles heaviest first; False: row-major.  Same gradients, bit for bit.
    Returns the previous setting."""
    return bool(_lib.r3dgs_set_tile_order(int(bool(on))))


def tile_order():
    return bool(_lib.r3dgs_set_tile_order(-1))


def set_bwd_segments(on):
    """True (default): the backward blend walks a long tile list -- at least max(2 S, 75 % of the pass's mean list length)
    entries, S = 128 (R3DGS_BWD_SEG_LEN, R3DGS_BWD_SEG_FACTOR; include/r3dgs_rasterizer.h) -- in segments of S entries, several
    workgroups per tile, from checkpoints the forward leaves; False: one workgroup per tile.  Gradients agree to rounding (the state at a
    segment's end is the forward's running product instead of the backward's own division chain).  A forward issued while
    this is off leaves no checkpoints, and its backward never splits.  Returns the previous setting."""
    return bool(_lib.r3dgs_set_bwd_segments(int(bool(on))))


def bwd_segments():
    return bool(_lib.r3dgs_set_bwd_segments(-1))


def export_tile_order(H, W, imageBuffer, P=0, num_rendered=0, binningBuffer=None):
    """Debug accessor (not in the reference): quad_depth int32[tiles, 4] (deepest contributor of each 8x8 quadrant, left by
    the forward) and -- given the binning buffer, P and num_rendered of the pass -- the launch order of the last backward
    over this state: `lists`, eight dicts (tile, segment, segments int64 arrays, each list heaviest first, and `walk`, the
    entries per list segment its units walk, 0 = whole tiles), and `units`, their concatenation."""
    n = ((W + 15) // 16) * ((H + 15) // 16)
    qd = torch.empty((n, 4), dtype=torch.int32, device=imageBuffer.device)
    out = dict(quad_depth=qd)
    order, R = None, 0
    if binningBuffer is not None and binningBuffer.numel():
        R = (num_rendered.capacity if isinstance(num_rendered, NumRendered) else
             _lib.r3dgs_binning_capacity(int(P), W, H, int(binningBuffer.numel())))
        cap = int(_lib.r3dgs_bwd_units_cap(R, W, H))
        order = torch.empty((cap + 2 * _ORDER_LISTS,), dtype=torch.int32, device=imageBuffer.device)
    with _on_device(imageBuffer.device):
        _check(_lib.r3dgs_export_tile_order(int(P), int(R), W, H, _ptr(binningBuffer) if order is not None else None,
                                            _ptr(imageBuffer), _ptr(qd), _ptr(order) if order is not None else None,
                                            _stream()), "export_tile_order")
    if order is not None:
        o = order.cpu().numpy().astype("int64") & 0xFFFFFFFF
        per = cap // _ORDER_LISTS
        lists = []
        for g in range(_ORDER_LISTS):   # list g: made from the 4 x 4 tile blocks g, g + 8, ...; consumed as entries b // 8 of list b % 8
            cnt, walk_log2 = int(o[cap + 2 * g]), int(o[cap + 2 * g + 1])
            u = o[g * per:g * per + cnt]
            lists.append(dict(tile=u & 0xFFFFF, segment=(u >> 20) & 63, segments=u >> 26, walk=(1 << walk_log2) if walk_log2 else 0))
        out["lists"] = lists
        out["units"] = {k: np.concatenate([l[k] for l in lists]) for k in ("tile", "segment", "segments")}
    return out


_ORDER_LISTS = 8


def rasterize_gaussians_counters(*args):
    """Forward in counter mode (calculate_mean_transmittance, forward.cu:560-564): same arguments as
    rasterize_gaussians; additionally returns (touched_pixels int32[P], transmittance fp32[P]).
    This is the building block of calculate_colours_variance (reduced_3dgs.cu:89-140)."""
    means3D = args[1]
    P = int(means3D.size(0))
    touched = torch.zeros((P,), dtype=torch.int32, device=means3D.device)
    transm = torch.zeros((P,), dtype=torch.float32, device=means3D.device)
    out = _forward_common(None, *args, counters=(touched, transm))
    return out + (touched, transm)


def calculate_colours_variance_partial(cam_positions, means3D, opacity, scales, rotations, cam_viewmatrices,
                                       cam_projmatrices, tan_fovxs, tan_fovys, image_height, image_width, sh, degrees,
                                       max_sh_deg):
    """The accumulation part of Reduced3DGS::calculateColourVariance (reduced_3dgs.cu:41-198) over the given cameras,
    WITHOUT the final divisions: (accum[P,max_sh_deg], wSum[P,1], mean[P,1,3], S[P,1,3]).  A camera-sharded run calls
    this per rank and merges the partials (multiview.merge_colour_variance)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P = int(means3D.size(0))
    D = int(max_sh_deg)
    opts = dict(dtype=torch.float32, device=dev)
    accum = torch.zeros((P, D), **opts)
    wSum, wSumSq = torch.zeros((P, 1), **opts), torch.zeros((P, 1), **opts)
    mean, variance = torch.zeros((P, 1, 3), **opts), torch.zeros((P, 1, 3), **opts)
    if P:
        m3, shc, deg = _dev_f32(means3D, dev), _dev_f32(sh, dev), _dev_i32(degrees, dev)
        M = int(shc.size(1))
        cams = _dev_f32(cam_positions, dev)
        Hs, Ws = image_height.tolist(), image_width.tolist()
        txs, tys = tan_fovxs.tolist(), tan_fovys.tolist()
        bg = torch.zeros(3, **opts)
        empty = torch.Tensor([])
        for i in range(int(cams.size(0)) if cams is not None else 0):
            touched = torch.zeros((P,), dtype=torch.int32, device=dev)
            transm = torch.zeros((P,), **opts)
            out = _forward_common(None, bg, m3, empty, opacity, scales, rotations, 1.0, empty, cam_viewmatrices[i],
                                  cam_projmatrices[i], txs[i], tys[i], Hs[i], Ws[i], shc, deg, cams[i], False, False,
                                  counters=(touched, transm))
            radii = out[2]
            with _on_device(dev):
                _check(_lib.r3dgs_colour_variance_accumulate(P, _ptr(deg), M, D, _ptr(m3), cams[i].data_ptr(),
                                                             _ptr(shc), _ptr(radii), _ptr(touched), _ptr(transm),
                                                             _ptr(wSum), _ptr(wSumSq), _ptr(mean), _ptr(variance),
                                                             _ptr(accum), _stream()), "calculate_colours_variance")
    return accum, wSum, mean, variance


def calculate_colours_variance(cam_positions, means3D, opacity, scales, rotations, cam_viewmatrices, cam_projmatrices,
                               tan_fovxs, tan_fovys, image_height, image_width, sh, degrees, max_sh_deg):
    """Reduced3DGS::calculateColourVariance (reduced_3dgs.cu:41-203) ->
    (colourDistances[P,max_sh_deg], variance[P,1,3], mean[P,1,3]).  Per camera: the counter-mode forward
    (touched pixels + summed transmittance per Gaussian) followed by ONE fused per-Gaussian accumulate kernel,
    instead of the reference's ~25 small torch operators per camera.  The four per-camera parameter tensors are
    read back once (the reference does a blocking .item() per value and camera)."""
    accum, wSum, mean, variance = calculate_colours_variance_partial(
        cam_positions, means3D, opacity, scales, rotations, cam_viewmatrices, cam_projmatrices, tan_fovxs, tan_fovys,
        image_height, image_width, sh, degrees, max_sh_deg)
    return accum / wSum, variance / wSum.view(-1, 1, 1), mean


def _dev_u8(t, dev):
    if t.device != dev:
        raise RuntimeError("all tensors must live on the same GPU")
    if t.dtype == torch.bool:
        t = t.view(torch.uint8) if t.is_contiguous() else t.contiguous().view(torch.uint8)
    if t.dtype != torch.uint8:
        raise RuntimeError("expected a bool tensor")
    return t.contiguous()


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensors must be on a GPU (no CPU path)")
    return t.device


def find_minimum_projected_pixel_size(w2ndc_transforms, w2ndc_transforms_inverse, means3D, image_height, image_width):
    """Reduced3DGS::calculatePixelSize (reduced_3dgs.cu:240-264) -> pixel_values[P,1] (10000 where unseen).
    One launch for all cameras; the image sizes stay on the device."""
    dev = _need_gpu(means3D, "find_minimum_projected_pixel_size")
    P = int(means3D.size(0))
    out = torch.empty((P, 1), dtype=torch.float32, device=dev)
    if P:
        m, mi = _dev_f32(w2ndc_transforms, dev), _dev_f32(w2ndc_transforms_inverse, dev)
        Hs, Ws = _dev_i32(image_height, dev), _dev_i32(image_width, dev)
        with _on_device(dev):
            _check(_lib.r3dgs_min_pixel_size(P, int(w2ndc_transforms.size(0)), _ptr(m), _ptr(mi), _ptr(_dev_f32(means3D, dev)),
                                             _ptr(Hs), _ptr(Ws), _ptr(out), _stream()),
                   "find_minimum_projected_pixel_size")
    return out


def sphere_ellipsoid_intersection(means3D, scales, rotations, neighbours_indices, sphere_radius, knn):
    """Reduced3DGS::intersectionTest (reduced_3dgs.cu:205-238) -> (redundancy_values int32[P,1],
    intersection_mask bool[P,knn])."""
    dev = _need_gpu(means3D, "sphere_ellipsoid_intersection")
    P, knn = int(means3D.size(0)), int(knn)
    red = torch.empty((P, 1), dtype=torch.int32, device=dev)
    mask = torch.empty((P, knn), dtype=torch.bool, device=dev)
    if P:
        nbr = _dev_i32(neighbours_indices, dev)
        if nbr.numel() != P * knn:
            raise RuntimeError("neighbours_indices must hold P * knn entries")
        with _on_device(dev):
            _check(_lib.r3dgs_sphere_ellipsoid_intersection(
                P, knn, _ptr(_dev_f32(means3D, dev)), _ptr(_dev_f32(scales, dev)), _ptr(_dev_f32(rotations, dev)),
                _ptr(nbr), _ptr(_dev_f32(sphere_radius, dev)), _ptr(red), mask.data_ptr() if knn else None,
                _stream()), "sphere_ellipsoid_intersection")
    return red, mask


def allocate_minimum_redundancy_value(redundancy_values, neighbours_indices, intersection_mask, knn):
    """Reduced3DGS::assignFinalRedundancyValue (reduced_3dgs.cu:268-287) -> (minimum_redundancy_values int32[P,1],)."""
    dev = _need_gpu(redundancy_values, "allocate_minimum_redundancy_value")
    P, knn = int(redundancy_values.size(0)), int(knn)
    out = torch.empty((P, 1), dtype=torch.int32, device=dev)
    if P:
        nbr, msk = _dev_i32(neighbours_indices, dev), _dev_u8(intersection_mask, dev)
        if nbr.numel() != P * knn or msk.numel() != P * knn:
            raise RuntimeError("neighbours_indices / intersection_mask must hold P * knn entries")
        with _on_device(dev):
            _check(_lib.r3dgs_min_redundancy(P, knn, _ptr(_dev_i32(redundancy_values, dev)), _ptr(nbr), _ptr(msk),
                                             _ptr(out), _stream()), "allocate_minimum_redundancy_value")
    return (out,)


def kmeans_cuda(values, centers, tol, max_iterations, _want_iterations=False):
    """Reduced3DGS::kmeans (reduced_3dgs.cu:290-340) -> (ids int32[n,1], centers fp32[n_centers]).  Values sorted
    once, one binary-search pass per update; the whole Lloyd loop is enqueued at once with the convergence flag on
    the device (the reference scans all centres per value and reads the shift back every iteration)."""
    dev = _need_gpu(values, "kmeans_cuda")
    n, nc = int(values.size(0)), int(centers.size(0))
    ids = torch.zeros((n, 1), dtype=torch.int32, device=dev)
    new_centers = torch.empty((nc,), dtype=torch.float32, device=dev)
    iters = torch.zeros((1,), dtype=torch.int32, device=dev)
    with _on_device(dev):
        ws_bytes = _lib.r3dgs_kmeans_workspace_bytes(n, nc)
        if ws_bytes == 0:
            raise RuntimeError(f"kmeans_cuda: {_lib.r3dgs_last_error().decode()}")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _check(_lib.r3dgs_kmeans(n, nc, _ptr(_dev_f32(values, dev)), _ptr(_dev_f32(centers, dev)), float(tol),
                                 int(max_iterations), _ptr(ids), _ptr(new_centers), _ptr(iters), _ptr(ws),
                                 _stream()), "kmeans_cuda")
    if _want_iterations:
        return ids, new_centers, iters
    return ids, new_centers


def pack_view_stats(viewspace_grad, radii, grad_norm_out, visible_out, radii_out):
    """One launch for the per-view densification statistics of a view-parallel step (include/r3dgs_reduction.h
    r3dgs_pack_view_stats): ||viewspace_grad[:, :2]|| where radii > 0, the visibility indicator and a copy of radii,
    written into caller-provided (exchange-buffer) tensors."""
    dev = _need_gpu(viewspace_grad, "pack_view_stats")
    P = int(radii.numel())
    if P == 0:
        return
    vg, rd = _dev_f32(viewspace_grad, dev), _dev_i32(radii, dev)
    for t, dt in ((grad_norm_out, torch.float32), (visible_out, torch.float32), (radii_out, torch.int32)):
        if t.device != dev or t.dtype != dt or t.numel() != P or not t.is_contiguous():
            raise RuntimeError("pack_view_stats: outputs must be contiguous [P] tensors on the same GPU")
    with _on_device(dev):
        _check(_lib.r3dgs_pack_view_stats(P, _ptr(vg), _ptr(rd), _ptr(grad_norm_out), _ptr(visible_out), _ptr(radii_out),
                                          _stream()), "pack_view_stats")


def reduce_shards(recv, world, shard_begin, sum_len, out):
    """Local half of the view-parallel exchange (include/r3dgs_reduction.h r3dgs_reduce_shards): recv [world, shard]
    fp32 buffer whose tail (global element index >= sum_len) holds int32 radii -> out [shard]."""
    dev = _need_gpu(recv, "reduce_shards")
    shard = int(out.numel())
    if recv.numel() != world * shard or not recv.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("reduce_shards: recv must be a contiguous [world, shard] buffer")
    with _on_device(dev):
        _check(_lib.r3dgs_reduce_shards(int(world), shard, int(shard_begin), int(sum_len), recv.data_ptr(),
                                        out.data_ptr(), _stream()), "reduce_shards")


def reduce_shards_mixed(recv, world, shard_begin, sum_len, half_end, out):
    """r3dgs_reduce_shards_mixed (include/r3dgs_reduction.h): as reduce_shards, with the words [sum_len, half_end) of the
    buffer holding bfloat16 pairs (fp32 accumulation in rank order, one rounding to nearest even)."""
    dev = _need_gpu(recv, "reduce_shards_mixed")
    shard = int(out.numel())
    if recv.numel() != world * shard or not recv.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("reduce_shards_mixed: recv must be a contiguous [world, shard] buffer")
    with _on_device(dev):
        _check(_lib.r3dgs_reduce_shards_mixed(int(world), shard, int(shard_begin), int(sum_len), int(half_end),
                                              recv.data_ptr(), out.data_ptr(), _stream()), "reduce_shards_mixed")
