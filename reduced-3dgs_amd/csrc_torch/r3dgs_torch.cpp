// r3dgs_torch.cpp -- compiled torch binding of the rasterizer's hot calls (gfx950 / PyTorch-ROCm).
//
// The reference binds its rasterizer through a torch C++ extension (DGR/ext.cpp:16-25, rasterize_points.cu:136-305:
// tensors in, tensors out, resizable byte tensors behind the allocator callbacks).  This file is that layer for the MI355X
// library: a pybind module over the C ABI of include/r3dgs_rasterizer.h (libr3dgs_hip.so) -- it moves pointers, allocates
// the outputs and the three state blobs with at::empty on the current HIP stream's device, and calls the library.  There
// is no arithmetic here and no other path.  The module does not link against the library: _C.py hands it the addresses of
// the entry points of the libr3dgs_hip.so IT loaded (bind()), so both layers drive one library instance -- one graph
// cache, one reservation advisor -- whichever build R3DGS_LIB selected; an unbound module refuses every call.
//
// Scope: the two calls a training step makes -- the asynchronous forward (r3dgs_forward_reserved + the strict-mode check
// of the pass header) and the backward.  Everything else (exact-size path, ragged inference forward, counter mode, the
// reduction operators, debug accessors) stays in the ctypes module diff_gaussian_rasterization/_C.py, which calls this
// one when it is built (R3DGS_BINDING=ctypes forces the pure-ctypes route).  Why it exists: at small scenes the step is
// host-bound and the ctypes marshalling of ~35 arguments per call is a third of it.
//
// Built by reduced-3dgs_amd/build.py with the host compiler against torch's headers (no hipify pass: the HIP-named c10
// headers are used directly).
#include <torch/extension.h>

// PyTorch-ROCm presents its HIP devices under the device type "cuda": the guard and stream accessors that accept that
// type are the *MasqueradingAsCUDA ones (the plain c10::hip::HIPGuard insists on DeviceType::HIP and throws)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>

#include "r3dgs_rasterizer.h"

namespace {

using at::Tensor;

// entry points of the loaded libr3dgs_hip.so (types taken from the C header)
struct Api {
#define R3_FN(name) decltype(&::name) name = nullptr;
    R3_FN(r3dgs_last_error)
    R3_FN(r3dgs_version)
    R3_FN(r3dgs_geometry_bytes)
    R3_FN(r3dgs_geometry_bytes_lean)
    R3_FN(r3dgs_binning_bytes)
    R3_FN(r3dgs_image_bytes)
    R3_FN(r3dgs_forward_hint)
    R3_FN(r3dgs_reserve_hint_view)
    R3_FN(r3dgs_forward_reserved)
    R3_FN(r3dgs_pass_query)
    R3_FN(r3dgs_backward)
    R3_FN(r3dgs_mark_visible)
#undef R3_FN
    bool bound = false;
} api;

void bind(const std::map<std::string, uintptr_t>& addr)
{
#define R3_FN(name)                                                                         \
    {                                                                                       \
        auto it = addr.find(#name);                                                         \
        if (it == addr.end() || !it->second) throw std::runtime_error("bind: no " #name);   \
        api.name = reinterpret_cast<decltype(api.name)>(it->second);                        \
    }
    R3_FN(r3dgs_last_error)
    R3_FN(r3dgs_version)
    R3_FN(r3dgs_geometry_bytes)
    R3_FN(r3dgs_geometry_bytes_lean)
    R3_FN(r3dgs_binning_bytes)
    R3_FN(r3dgs_image_bytes)
    R3_FN(r3dgs_forward_hint)
    R3_FN(r3dgs_reserve_hint_view)
    R3_FN(r3dgs_forward_reserved)
    R3_FN(r3dgs_pass_query)
    R3_FN(r3dgs_backward)
    R3_FN(r3dgs_mark_visible)
#undef R3_FN
    api.bound = true;
}

void need_bound()
{
    if (!api.bound) throw std::runtime_error("r3dgs torch binding: not bound to libr3dgs_hip.so (import diff_gaussian_rasterization._C)");
}

[[noreturn]] void fail(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + api.r3dgs_last_error());
}

// absent optional input: an empty tensor, as the reference's wrapper passes torch.Tensor([])
// (diff_gaussian_rasterization/__init__.py:209-218)
template <class T>
const T* opt_ptr(const Tensor& t)
{
    return t.defined() && t.numel() != 0 ? t.data_ptr<T>() : nullptr;
}

Tensor dev_f32(const Tensor& t, const c10::Device& dev)
{
    if (!t.defined() || t.numel() == 0) return Tensor();
    if (t.device() != dev) throw std::runtime_error("expected a tensor on " + dev.str() + ", got " + t.device().str());
    if (t.scalar_type() != at::kFloat) throw std::runtime_error(std::string("expected float32, got ") + c10::toString(t.scalar_type()));
    return t.contiguous();
}

Tensor dev_i32(const Tensor& t, const c10::Device& dev)
{
    if (!t.defined() || t.numel() == 0) return Tensor();
    if (t.device() != dev) throw std::runtime_error("expected a tensor on " + dev.str() + ", got " + t.device().str());
    if (t.scalar_type() != at::kInt) throw std::runtime_error(std::string("expected int32, got ") + c10::toString(t.scalar_type()));
    return t.contiguous();
}

std::mutex g_mu;
std::map<std::tuple<int, int, int, int, int>, size_t> g_sizes;   // (kind, a, b, c, d) -> bytes

size_t blob_bytes(int kind, int a, int b = 0, int c = 0, int d = 0)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const auto key = std::make_tuple(kind, a, b, c, d);
    auto it = g_sizes.find(key);
    if (it != g_sizes.end()) return it->second;
    size_t v = 0;
    switch (kind) {
        case 0: v = api.r3dgs_geometry_bytes(a); break;
        case 1: v = api.r3dgs_geometry_bytes_lean(a); break;
        case 2: v = api.r3dgs_binning_bytes(a, b, c, d); break;
        default: v = api.r3dgs_image_bytes(a, b); break;
    }
    if (v == 0) fail("rasterize_gaussians");
    if (g_sizes.size() > 4096) g_sizes.clear();
    g_sizes[key] = v;
    return v;
}

// RasterizeGaussiansCUDA (rasterize_points.cu:136-222) on the asynchronous path.
// -> (ticket, reserve, num_rendered, flags, out_color, radii, geom, binning, img).  ticket == 0: nothing is known about
// this view size yet -- the caller takes the exact-size path.  num_rendered / flags are filled (>= 0) when `strict`: the
// call then waited for the pass's HEADER (not for the pass), with the GIL released.
std::tuple<long long, int, int, int, Tensor, Tensor, Tensor, Tensor, Tensor> forward_reserved(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp, const Tensor& viewmatrix,
    const Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, const Tensor& sh,
    const Tensor& degrees, const Tensor& campos, bool prefiltered, bool trains, bool strict)
{
    need_bound();
    if (means3D.dim() != 2 || means3D.size(1) != 3)
        throw std::runtime_error("means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:158-161
    const c10::Device dev = means3D.device();
    if (!dev.is_cuda()) throw std::runtime_error("the MI355X rasterizer needs device tensors (no CPU path)");
    const int P = (int)means3D.size(0), H = (int)image_height, W = (int)image_width;
    const Tensor bg = dev_f32(background, dev), m3 = dev_f32(means3D, dev), col = dev_f32(colors, dev);
    const Tensor op = dev_f32(opacity, dev), sc = dev_f32(scales, dev), rot = dev_f32(rotations, dev);
    const Tensor cov = dev_f32(cov3D_precomp, dev), vm = dev_f32(viewmatrix, dev), pm = dev_f32(projmatrix, dev);
    const Tensor cp = dev_f32(campos, dev), shc = dev_f32(sh, dev), deg = dev_i32(degrees, dev);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    api.r3dgs_forward_hint(trains ? 1 : 0);
    const int reserve = api.r3dgs_reserve_hint_view(P, W, H, opt_ptr<float>(vm));
    if (reserve < 0) fail("rasterize_gaussians");
    Tensor none;
    if (reserve == 0) return {0LL, 0, -1, 0, none, none, none, none, none};
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev), i32 = f32.dtype(at::kInt), u8 = f32.dtype(at::kByte);
    Tensor out_color = at::empty({3, H, W}, f32), radii = at::empty({P}, i32);
    const bool lean = !trains || !shc.defined() || col.defined();   // no SH direction derivatives will be left
    Tensor geom = at::empty({(int64_t)blob_bytes(lean ? 1 : 0, P)}, u8);
    Tensor binning = at::empty({(int64_t)blob_bytes(2, P, W, H, reserve)}, u8);
    Tensor img = at::empty({(int64_t)blob_bytes(3, W, H)}, u8);
    void* stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    const int M = shc.defined() ? (int)shc.size(1) : 0;
    const long long ticket = api.r3dgs_forward_reserved(
        reinterpret_cast<char*>(geom.data_ptr()), reinterpret_cast<char*>(binning.data_ptr()),
        reinterpret_cast<char*>(img.data_ptr()), reserve, P, opt_ptr<int>(deg), M, opt_ptr<float>(bg), W, H, opt_ptr<float>(m3),
        opt_ptr<float>(shc), opt_ptr<float>(col), opt_ptr<float>(op), opt_ptr<float>(sc), (float)scale_modifier,
        opt_ptr<float>(rot), opt_ptr<float>(cov), opt_ptr<float>(vm), opt_ptr<float>(pm), opt_ptr<float>(cp), (float)tan_fovx,
        (float)tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(), nullptr, nullptr, radii.data_ptr<int>(), 0, 0, stream);
    if (ticket < 0) fail("rasterize_gaussians");
    int rendered = -1, flags = 0;
    if (strict && ticket > 0) {
        int visible = 0, cap = 0, st;
        {
            pybind11::gil_scoped_release nogil;   // a poll of host memory with short sleeps inside the library
            st = api.r3dgs_pass_query(ticket, 1, &rendered, &visible, &cap, &flags);
        }
        if (st < 0) fail("num_rendered");
        if (st != 1) rendered = -1;
    }
    return {ticket, reserve, rendered, flags, out_color, radii, geom, binning, img};
}

// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:224-305).  `capacity`: the pair capacity the forward carved the
// binning blob with.  Every element of every output is written by the library: at::empty, no fills.
std::vector<Tensor> backward(const Tensor& background, const Tensor& means3D, const Tensor& radii, const Tensor& colors,
                             const Tensor& scales, const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp,
                             const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy,
                             const Tensor& dL_dout_color, const Tensor& sh, const Tensor& degrees, const Tensor& campos,
                             const Tensor& geomBuffer, int64_t capacity, const Tensor& binningBuffer, const Tensor& imageBuffer,
                             double lambda_sh_sparsity, bool debug, bool want_conic)
{
    need_bound();
    const c10::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const int M = (sh.defined() && sh.numel() != 0) ? (int)sh.size(1) : 0;
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    if (P == 0) {
        return {at::zeros({0, 3}, f32), at::zeros({0, 3}, f32), at::zeros({0, 1}, f32), at::zeros({0, 3}, f32),
                at::zeros({0, 6}, f32), at::zeros({0, M, 3}, f32), at::zeros({0, 3}, f32), at::zeros({0, 4}, f32)};
    }
    Tensor dL_dmeans3D = at::empty({P, 3}, f32), dL_dmeans2D = at::empty({P, 3}, f32), dL_dcolors = at::empty({P, 3}, f32);
    Tensor dL_dopacity = at::empty({P, 1}, f32), dL_dcov3D = at::empty({P, 6}, f32), dL_dsh = at::empty({P, M, 3}, f32);
    Tensor dL_dscales = at::empty({P, 3}, f32), dL_drotations = at::empty({P, 4}, f32);
    Tensor dL_dconic = want_conic ? at::empty({P, 2, 2}, f32) : Tensor();
    const Tensor bg = dev_f32(background, dev), m3 = dev_f32(means3D, dev), col = dev_f32(colors, dev);
    const Tensor sc = dev_f32(scales, dev), rot = dev_f32(rotations, dev), cov = dev_f32(cov3D_precomp, dev);
    const Tensor vm = dev_f32(viewmatrix, dev), pm = dev_f32(projmatrix, dev), cp = dev_f32(campos, dev);
    const Tensor g = dev_f32(dL_dout_color, dev), shc = dev_f32(sh, dev), deg = dev_i32(degrees, dev), rad = dev_i32(radii, dev);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    void* stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    auto blob = [](const Tensor& t) { return t.defined() && t.numel() != 0 ? reinterpret_cast<char*>(t.data_ptr()) : nullptr; };
    const int st = api.r3dgs_backward(
        P, opt_ptr<int>(deg), M, (int)capacity, opt_ptr<float>(bg), W, H, opt_ptr<float>(m3), opt_ptr<float>(shc),
        opt_ptr<float>(col), opt_ptr<float>(sc), (float)scale_modifier, opt_ptr<float>(rot), opt_ptr<float>(cov),
        opt_ptr<float>(vm), opt_ptr<float>(pm), opt_ptr<float>(cp), (float)tan_fovx, (float)tan_fovy, opt_ptr<int>(rad),
        blob(geomBuffer), blob(binningBuffer), blob(imageBuffer), opt_ptr<float>(g), dL_dmeans2D.data_ptr<float>(),
        dL_dconic.defined() ? dL_dconic.data_ptr<float>() : nullptr, dL_dopacity.data_ptr<float>(),
        dL_dcolors.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(),
        M ? dL_dsh.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>(),
        (float)lambda_sh_sparsity, debug ? 1 : 0, stream);
    if (st < 0) fail("rasterize_gaussians_backward");
    std::vector<Tensor> out{dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations};
    if (want_conic) out.push_back(dL_dconic);
    return out;
}

// markVisible (rasterize_points.cu:307-326)
Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix)
{
    need_bound();
    const c10::Device dev = means3D.device();
    const int P = (int)means3D.size(0);
    Tensor present = at::zeros({P}, at::TensorOptions().dtype(at::kBool).device(dev));
    if (P) {
        const Tensor m3 = dev_f32(means3D, dev), vm = dev_f32(viewmatrix, dev), pm = dev_f32(projmatrix, dev);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        if (api.r3dgs_mark_visible(P, opt_ptr<float>(m3), opt_ptr<float>(vm), opt_ptr<float>(pm),
                               reinterpret_cast<unsigned char*>(present.data_ptr()),
                               c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream()) < 0)
            fail("mark_visible");
    }
    return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "compiled torch binding of libr3dgs_hip.so's hot calls (see diff_gaussian_rasterization/_C.py)";
    m.def("bind", &bind);
    m.def("forward_reserved", &forward_reserved);
    m.def("backward", &backward);
    m.def("mark_visible", &mark_visible);
    m.def("library_version", []() {
        need_bound();
        return std::string(api.r3dgs_version());
    });
}
