// A COMPILED torch-extension module `_C` over the C ABI of include/gsicp_hip.h: the pybind surface the reference's two torch extensions
// present to their Python packages, for a maintainer who wants `diff_gaussian_rasterization._C` and `simple_knn._C` to be extension
// modules the way the upstream submodules build them (the submodules are empty in the reference tree; the surface is the one SURVEY
// section 8(b) lists: `rasterize_gaussians`, `rasterize_gaussians_backward`, `mark_visible` behind
// [REF gaussian_renderer/__init__.py:14, 259, 294-302] and `distCUDA2` behind [REF scene/gaussian_model.py:20]).
//   * torch tensors in, torch tensors out; everything runs on torch's CURRENT stream of the tensors' device;
//   * the three scratch buffers that live from the forward to the backward are torch byte tensors grown through resize callbacks
//     [REF SIBR_viewers/src/projects/gaussianviewer/renderer/GaussianView.cpp:304, 432-434] and returned to Python, which saves them;
//   * this fork's two extra outputs (depth image, is_used) ride along [REF gaussian_renderer/__init__.py:294-302];
//   * no arithmetic lives here: every function is argument checking, output allocation and ONE C-ABI call.
// One source, one module name (`_C`); gs_icp_slam_amd/build.py compiles it once and places a copy in each of the two packages.
// The library is dlopen'ed relative to the packages (integration/torch_ext/<package>/_C.so -> gs_icp_slam_amd/libgsicp_hip.so), never linked: it must bind to the HIP runtime that
// libtorch (a link-time dependency of this module) has already brought into the process.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <tuple>

#include "../include/gsicp_hip.h"

namespace py = pybind11;

namespace {
#define GSICP_API(X) \
    X(gsicp_abi_version) \
    X(gsicp_knn_dist2) \
    X(gsicp_last_error) \
    X(gsicp_raster_backward) \
    X(gsicp_raster_backward_scratch_bytes) \
    X(gsicp_raster_forward) \
    X(gsicp_raster_mark_visible)
struct Api {
#define X(name) decltype(&::name) name = nullptr;
    GSICP_API(X)
#undef X
    bool loaded = false;
} A;

void load_library() {
    if (A.loaded) return;
    Dl_info info;
    if (!dladdr((void*)&load_library, &info) || !info.dli_fname) throw std::runtime_error("_C: cannot locate the module file");
    std::string dir(info.dli_fname);
    dir = dir.substr(0, dir.find_last_of('/'));
    const std::string path = dir + "/../../../gs_icp_slam_amd/libgsicp_hip.so";   // integration/torch_ext/<package>/_C.so -> <repo>/gs_icp_slam_amd/
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw std::runtime_error("_C: " + path + " not loadable (build it with `python -m gs_icp_slam_amd.build`; there is no CPU path): " + dlerror());
#define X(name) A.name = (decltype(&::name))dlsym(h, #name); if (!A.name) throw std::runtime_error("_C: libgsicp_hip.so lacks " #name);
    GSICP_API(X)
#undef X
    if (A.gsicp_abi_version() != GSICP_ABI_VERSION)
        throw std::runtime_error("_C: libgsicp_hip.so has ABI " + std::to_string(A.gsicp_abi_version()) + ", this module was built for " + std::to_string(GSICP_ABI_VERSION));
    A.loaded = true;
    if (std::getenv("GSICP_ANNOUNCE")) {
        char real[4096];
        py::print("GSICP_LOADED", realpath(path.c_str(), real) ? real : path.c_str(), "pid=" + std::to_string((long)getpid()), "via=compiled-torch-ext",
                  py::arg("flush") = true);
    }
}

void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + A.gsicp_last_error());
}

// the library's resize convention: user = the torch byte tensor to grow
char* resize_byte_tensor(void* user, size_t bytes) {
    torch::Tensor& t = *reinterpret_cast<torch::Tensor*>(user);
    t.resize_({(long long)(bytes > 0 ? bytes : 1)});
    return reinterpret_cast<char*>(t.data_ptr());
}

torch::Tensor f32c(const torch::Tensor& t, const torch::Device& dev) {
    if (!t.defined() || t.numel() == 0) return torch::Tensor();
    return t.to(torch::TensorOptions().device(dev).dtype(torch::kFloat32)).contiguous();
}
const float* fptr(const torch::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(torch::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

void* current_stream(const torch::Device& dev) { return (void*)c10::hip::getCurrentHIPStream(dev.index()).stream(); }

void need_device(const torch::Tensor& t, const char* who) {
    if (!t.is_cuda()) throw std::runtime_error(std::string(who) + " (gfx950): tensors must live on the HIP device; there is no CPU path");
}

// (num_rendered, colour (3,H,W), depth (1,H,W), radii (P) int32, is_used (P) int32, geomBuffer, binningBuffer, imgBuffer)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                    const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                    const int image_height, const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                    const bool prefiltered, const bool debug) {
    need_device(means3D, "rasterize_gaussians");
    load_library();
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    const torch::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    const torch::Tensor m3 = f32c(means3D, dev), col = f32c(colors, dev), op = f32c(opacity, dev), sc = f32c(scales, dev), rot = f32c(rotations, dev),
                        cov = f32c(cov3D_precomp, dev), shs = f32c(sh, dev), bg = f32c(background, dev), view = f32c(viewmatrix, dev),
                        proj = f32c(projmatrix, dev), cam = f32c(campos, dev);
    const int M = shs.defined() ? (shs.dim() == 3 ? (int)shs.size(1) : (int)(shs.numel() / (3 * (P > 0 ? P : 1)))) : 0;
    const auto f32 = torch::TensorOptions().device(dev).dtype(torch::kFloat32);
    const auto i32 = torch::TensorOptions().device(dev).dtype(torch::kInt32);
    const auto u8 = torch::TensorOptions().device(dev).dtype(torch::kUInt8);
    torch::Tensor out_color = torch::empty({3, H, W}, f32), out_depth = torch::empty({1, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, i32), is_used = torch::empty({P}, i32);
    torch::Tensor geom = torch::empty({0}, u8), binning = torch::empty({0}, u8), img = torch::empty({0}, u8);
    const int rendered = A.gsicp_raster_forward(resize_byte_tensor, &geom, resize_byte_tensor, &binning, resize_byte_tensor, &img, P, degree, M, fptr(bg), W, H,
                                                fptr(m3), fptr(shs), fptr(col), fptr(op), fptr(sc), scale_modifier, fptr(rot), fptr(cov), fptr(view),
                                                fptr(proj), fptr(cam), tan_fovx, tan_fovy, prefiltered ? 1 : 0, fptr_mut(out_color), fptr_mut(out_depth),
                                                radii.data_ptr<int>(), is_used.data_ptr<int>(), 1, 0, debug ? 1 : 0, 0, current_stream(dev));
    check(rendered, "rasterize_gaussians");
    return std::make_tuple(rendered, out_color, out_depth, radii, is_used, geom, binning, img);
}

// (dL_dmeans2D (P,3), dL_dcolors (P,3), dL_dopacity (P,1), dL_dmeans3D (P,3), dL_dcov3D (P,6), dL_dsh (P,M,3), dL_dscales (P,3), dL_drotations (P,4));
// an output whose input was not given is an empty tensor
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
                             const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                             const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                             const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_depth, const torch::Tensor& sh, const int degree,
                             const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                             const torch::Tensor& imageBuffer, const bool debug) {
    need_device(means3D, "rasterize_gaussians_backward");
    load_library();
    const torch::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    if (!dL_dout_color.defined() || dL_dout_color.dim() != 3) throw std::runtime_error("dL_dout_color must have dimensions (3, H, W)");
    const int P = (int)means3D.size(0), H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const torch::Tensor m3 = f32c(means3D, dev), col = f32c(colors, dev), sc = f32c(scales, dev), rot = f32c(rotations, dev), cov = f32c(cov3D_precomp, dev),
                        shs = f32c(sh, dev), bg = f32c(background, dev), view = f32c(viewmatrix, dev), proj = f32c(projmatrix, dev), cam = f32c(campos, dev),
                        g_color = f32c(dL_dout_color, dev), g_depth = f32c(dL_dout_depth, dev);
    const int M = shs.defined() ? (shs.dim() == 3 ? (int)shs.size(1) : (int)(shs.numel() / (3 * (P > 0 ? P : 1)))) : 0;
    const auto f32 = torch::TensorOptions().device(dev).dtype(torch::kFloat32);
    const auto u8 = torch::TensorOptions().device(dev).dtype(torch::kUInt8);
    torch::Tensor dL_dmeans2D = torch::empty({P, 3}, f32), dL_dopacity = torch::empty({P, 1}, f32), dL_dmeans3D = torch::empty({P, 3}, f32);
    torch::Tensor dL_dcolors = col.defined() ? torch::empty({P, 3}, f32) : torch::Tensor();
    torch::Tensor dL_dcov3D = cov.defined() ? torch::empty({P, 6}, f32) : torch::Tensor();
    torch::Tensor dL_dsh = shs.defined() ? torch::empty({P, M, 3}, f32) : torch::Tensor();
    torch::Tensor dL_dscales = sc.defined() ? torch::empty({P, 3}, f32) : torch::Tensor();
    torch::Tensor dL_drots = rot.defined() ? torch::empty({P, 4}, f32) : torch::Tensor();
    if (P > 0) {
        torch::Tensor scratch = torch::empty({(long long)A.gsicp_raster_backward_scratch_bytes(R, W, H)}, u8);
        const torch::Tensor rad = radii.to(torch::TensorOptions().device(dev).dtype(torch::kInt32)).contiguous();
        check(A.gsicp_raster_backward(P, degree, M, R, fptr(bg), W, H, fptr(m3), fptr(shs), fptr(col), fptr(sc), scale_modifier, fptr(rot), fptr(cov), fptr(view),
                                      fptr(proj), fptr(cam), tan_fovx, tan_fovy, rad.data_ptr<int>(), (const char*)geomBuffer.data_ptr(),
                                      (const char*)binningBuffer.data_ptr(), (const char*)imageBuffer.data_ptr(), (char*)scratch.data_ptr(), fptr(g_color),
                                      fptr(g_depth), fptr_mut(dL_dmeans2D), nullptr, fptr_mut(dL_dopacity), fptr_mut(dL_dcolors), nullptr,
                                      fptr_mut(dL_dmeans3D), fptr_mut(dL_dcov3D), fptr_mut(dL_dsh), fptr_mut(dL_dscales), fptr_mut(dL_drots), 1, 0,
                                      debug ? 1 : 0, 0, nullptr, nullptr, 0, current_stream(dev)),
              "rasterize_gaussians_backward");
    }
    const torch::Tensor none = torch::empty({0}, f32);
    auto or_none = [&](const torch::Tensor& t) { return t.defined() ? t : none; };
    return std::make_tuple(dL_dmeans2D, or_none(dL_dcolors), dL_dopacity, dL_dmeans3D, or_none(dL_dcov3D), or_none(dL_dsh), or_none(dL_dscales), or_none(dL_drots));
}

torch::Tensor mark_visible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix) {
    need_device(means3D, "mark_visible");
    load_library();
    const torch::Device dev = means3D.device();
    c10::DeviceGuard guard(dev);
    const int P = (int)means3D.size(0);
    const torch::Tensor m3 = f32c(means3D, dev), view = f32c(viewmatrix, dev), proj = f32c(projmatrix, dev);
    torch::Tensor present = torch::empty({P}, torch::TensorOptions().device(dev).dtype(torch::kUInt8));
    if (P > 0) check(A.gsicp_raster_mark_visible(P, fptr(m3), fptr(view), fptr(proj), present.data_ptr<unsigned char>(), current_stream(dev)), "mark_visible");
    return present.to(torch::kBool);
}

// simple_knn._C.distCUDA2: mean squared distance to the three nearest other points, (P,) f32
torch::Tensor distCUDA2(const torch::Tensor& points) {
    need_device(points, "simple_knn.distCUDA2");
    load_library();
    const torch::Device dev = points.device();
    c10::DeviceGuard guard(dev);
    const torch::Tensor pts = points.detach().to(torch::kFloat32).contiguous().view({-1, 3});
    const int P = (int)pts.size(0);
    torch::Tensor out = torch::empty({P}, torch::TensorOptions().device(dev).dtype(torch::kFloat32));
    if (P > 0) check(A.gsicp_knn_dist2(P, pts.data_ptr<float>(), out.data_ptr<float>(), current_stream(dev)), "simple_knn.distCUDA2");
    return out;
}
}  // namespace

PYBIND11_MODULE(_C, m) {
    m.doc() = "diff_gaussian_rasterization._C / simple_knn._C over libgsicp_hip.so (gfx950)";
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("mark_visible", &mark_visible);
    m.def("distCUDA2", &distCUDA2);
}
