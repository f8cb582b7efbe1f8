// A COMPILED `pygicp` module (PyInit_pygicp) over the C ABI of include/gsicp_hip.h — the binding of INTEGRATION.md §3.2 made real, for a
// maintainer who wants the reference's `import pygicp` [REF mp_Tracker.py:10] to resolve to an extension module the way fast_gicp's own
// pybind11 binding does (upstream: src/python/main.cpp).  It covers exactly the surface the reference's trackers call
// [REF mp_Tracker.py:53, 109-110, 157-169, 191-199, 231, 256-264, 287-288, 301-309] plus the three upstream setters used for CPU-baseline
// runs, takes and returns host numpy arrays, raises RuntimeError on failure, and pickles (the reference ships the Tracker — with its
// FastGICP inside — to a spawned process [REF gs_icp_slam.py:121-127]).  No arithmetic lives here: every method is one C-ABI call.
// The product's default binding is the ctypes mirror gs_icp_slam_amd/gicp.py (which adds the device-tensor overloads); this module is
// built by gs_icp_slam_amd/build.py into integration/pygicp.*.so and checked against the mirror by tests/test_pybind_module.py.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>

#include <map>
#include <stdexcept>
#include <string>

#include "../include/gsicp_hip.h"

namespace py = pybind11;

namespace {
// The library is opened at import, NOT linked: libgsicp_hip.so must run on the HIP runtime PyTorch carries when PyTorch is in the process
// (two runtimes in one process cannot both open the device), so the import order matters and is fixed here exactly as the ctypes mirror
// fixes it (gs_icp_slam_amd/_lib.py): import torch if it is installed — the reference imports it before pygicp anyway
// [REF mp_Tracker.py:2, 10] — then dlopen the library next to this module and bind the entry points by name.
#define GSICP_API(X) \
    X(gsicp_gicp_align) \
    X(gsicp_gicp_calculate_target_covariance_with_filter) \
    X(gsicp_gicp_create) \
    X(gsicp_gicp_destroy) \
    X(gsicp_gicp_get_source_correspondence) \
    X(gsicp_gicp_get_source_rotationsq) \
    X(gsicp_gicp_get_source_scales) \
    X(gsicp_gicp_get_target_rotationsq) \
    X(gsicp_gicp_get_target_scales) \
    X(gsicp_gicp_num_source) \
    X(gsicp_gicp_num_target) \
    X(gsicp_gicp_set_correspondence_randomness) \
    X(gsicp_gicp_set_input_source) \
    X(gsicp_gicp_set_input_target) \
    X(gsicp_gicp_set_max_correspondence_distance) \
    X(gsicp_gicp_set_max_iterations) \
    X(gsicp_gicp_set_max_knn_distance) \
    X(gsicp_gicp_set_num_threads) \
    X(gsicp_gicp_set_source_filter) \
    X(gsicp_gicp_set_target_covariances_fromqs) \
    X(gsicp_gicp_set_target_filter) \
    X(gsicp_last_error)
struct Api {
#define X(name) decltype(&::name) name = nullptr;
    GSICP_API(X)
#undef X
} A;

void load_library() {
    try { py::module_::import("torch"); } catch (py::error_already_set&) { PyErr_Clear(); }
    Dl_info info;
    if (!dladdr((void*)&load_library, &info) || !info.dli_fname) throw std::runtime_error("pygicp: cannot locate the module file");
    std::string dir(info.dli_fname);
    dir = dir.substr(0, dir.find_last_of('/'));
    const std::string path = dir + "/../gs_icp_slam_amd/libgsicp_hip.so";
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw std::runtime_error("pygicp: " + path + " not loadable (build it with `python -m gs_icp_slam_amd.build`; there is no CPU path): " + dlerror());
#define X(name) A.name = (decltype(&::name))dlsym(h, #name); if (!A.name) throw std::runtime_error("pygicp: libgsicp_hip.so lacks " #name);
    GSICP_API(X)
#undef X
    if (std::getenv("GSICP_ANNOUNCE")) {      // tools/run_reference_slam.py: which processes of a reference run loaded the library, and through what
        char real[4096];
        py::print("GSICP_LOADED", realpath(path.c_str(), real) ? real : path.c_str(), "pid=" + std::to_string((long)getpid()), "via=compiled-pygicp",
                  py::arg("flush") = true);
    }
}

void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string("pygicp.FastGICP.") + what + ": " + A.gsicp_last_error());
}

struct FastGICP {
    gsicp_gicp* h = nullptr;
    std::map<std::string, double> config;      // setter name -> last value: what pickling replays
    std::string regularization = "";
    FastGICP() {
        h = A.gsicp_gicp_create();
        if (!h) throw std::runtime_error(std::string("pygicp.FastGICP (gfx950): ") + A.gsicp_last_error());
    }
    ~FastGICP() { if (h) A.gsicp_gicp_destroy(h); }
    FastGICP(const FastGICP&) = delete;
    FastGICP& operator=(const FastGICP&) = delete;

    void set_max_correspondence_distance(double d) { config["max_corr"] = d; check(A.gsicp_gicp_set_max_correspondence_distance(h, d), "set_max_correspondence_distance"); }
    void set_max_knn_distance(double d) { config["max_knn"] = d; check(A.gsicp_gicp_set_max_knn_distance(h, d), "set_max_knn_distance"); }
    void set_correspondence_randomness(int k) { config["k"] = k; check(A.gsicp_gicp_set_correspondence_randomness(h, k), "set_correspondence_randomness"); }
    void set_max_iterations(int n) { config["iters"] = n; check(A.gsicp_gicp_set_max_iterations(h, n), "set_max_iterations"); }
    void set_num_threads(int n) { config["threads"] = n; check(A.gsicp_gicp_set_num_threads(h, n), "set_num_threads"); }

    // (N,3) float32 or float64, row-major [REF mp_Tracker.py:157 float64; :191, :287 float32]
    void set_cloud(const py::array& pts, bool target) {
        if (pts.ndim() != 2 || pts.shape(1) != 3) throw std::runtime_error("pygicp.FastGICP: points must have shape (N, 3)");
        if (pts.dtype().is(py::dtype::of<double>())) {
            auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(pts);
            check(target ? A.gsicp_gicp_set_input_target(h, a.data(), (int)a.shape(0), 1) : A.gsicp_gicp_set_input_source(h, a.data(), (int)a.shape(0), 1),
                  target ? "set_input_target" : "set_input_source");
        } else {
            auto a = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(pts);
            check(target ? A.gsicp_gicp_set_input_target(h, a.data(), (int)a.shape(0), 0) : A.gsicp_gicp_set_input_source(h, a.data(), (int)a.shape(0), 0),
                  target ? "set_input_target" : "set_input_source");
        }
    }
    void set_filter(int n_trackable, py::array_t<int, py::array::c_style | py::array::forcecast> f, bool target) {
        check(target ? A.gsicp_gicp_set_target_filter(h, n_trackable, f.data(), (int)f.size()) : A.gsicp_gicp_set_source_filter(h, n_trackable, f.data(), (int)f.size()),
              target ? "set_target_filter" : "set_source_filter");
    }
    py::array_t<float> fetch(int (*fn)(gsicp_gicp*, float*, int), int n, int width, const char* what) {
        py::array_t<float> out((size_t)n * width);
        const int got = fn(h, out.mutable_data(), n);
        check(got, what);
        out.resize({(size_t)got * width});
        return out;      // flat: the caller reshapes to (-1, 4) / (-1, 3) [REF mp_Tracker.py:168-169]
    }
    py::array_t<float> align(py::object initial_guess) {
        double init[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, out[16];
        if (!initial_guess.is_none()) {
            auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(initial_guess);
            if (!a || a.ndim() != 2 || a.shape(0) != 4 || a.shape(1) != 4) throw std::runtime_error("pygicp.FastGICP.align: initial guess must be 4x4");
            for (int i = 0; i < 16; ++i) init[i] = a.data()[i];
        }
        int rc;
        {
            py::gil_scoped_release nogil;       // the call waits for the registration kernel
            rc = A.gsicp_gicp_align(h, init, out);
        }
        check(rc, "align");
        py::array_t<float> T({4, 4});           // the reference binding returns an Eigen::Matrix4f
        for (int i = 0; i < 16; ++i) T.mutable_data()[i] = (float)out[i];
        return T;
    }
    py::tuple get_source_correspondence() {
        const int n = A.gsicp_gicp_num_source(h);
        py::array_t<int> idx((size_t)n);
        py::array_t<float> d2((size_t)n);
        const int got = A.gsicp_gicp_get_source_correspondence(h, idx.mutable_data(), d2.mutable_data(), n);
        check(got, "get_source_correspondence");
        idx.resize({(size_t)got}); d2.resize({(size_t)got});
        return py::make_tuple(idx, d2);
    }
};
}  // namespace

PYBIND11_MODULE(pygicp, m) {
    load_library();
    m.doc() = "GS-ICP-SLAM tracker (fast_gicp API) on MI355X: compiled binding over libgsicp_hip.so";
    py::class_<FastGICP>(m, "FastGICP")
        .def(py::init<>())
        .def("set_max_correspondence_distance", &FastGICP::set_max_correspondence_distance)
        .def("set_max_knn_distance", &FastGICP::set_max_knn_distance)
        .def("set_correspondence_randomness", &FastGICP::set_correspondence_randomness)
        .def("set_max_iterations", &FastGICP::set_max_iterations)
        .def("set_num_threads", &FastGICP::set_num_threads)
        .def("set_input_target", [](FastGICP& g, const py::array& p) { g.set_cloud(p, true); })
        .def("set_input_source", [](FastGICP& g, const py::array& p) { g.set_cloud(p, false); })
        .def("set_target_filter", [](FastGICP& g, int n, py::array_t<int, py::array::c_style | py::array::forcecast> f) { g.set_filter(n, f, true); })
        .def("set_source_filter", [](FastGICP& g, int n, py::array_t<int, py::array::c_style | py::array::forcecast> f) { g.set_filter(n, f, false); })
        .def("calculate_target_covariance_with_filter",
             [](FastGICP& g) { check(A.gsicp_gicp_calculate_target_covariance_with_filter(g.h), "calculate_target_covariance_with_filter"); })
        .def("get_target_rotationsq", [](FastGICP& g) { return g.fetch(A.gsicp_gicp_get_target_rotationsq, A.gsicp_gicp_num_target(g.h), 4, "get_target_rotationsq"); })
        .def("get_target_scales", [](FastGICP& g) { return g.fetch(A.gsicp_gicp_get_target_scales, A.gsicp_gicp_num_target(g.h), 3, "get_target_scales"); })
        .def("get_source_rotationsq", [](FastGICP& g) { return g.fetch(A.gsicp_gicp_get_source_rotationsq, A.gsicp_gicp_num_source(g.h), 4, "get_source_rotationsq"); })
        .def("get_source_scales", [](FastGICP& g) { return g.fetch(A.gsicp_gicp_get_source_scales, A.gsicp_gicp_num_source(g.h), 3, "get_source_scales"); })
        .def("set_target_covariances_fromqs",
             [](FastGICP& g, py::array_t<float, py::array::c_style | py::array::forcecast> r, py::array_t<float, py::array::c_style | py::array::forcecast> s) {
                 check(A.gsicp_gicp_set_target_covariances_fromqs(g.h, r.data(), (int)r.size(), s.data(), (int)s.size()), "set_target_covariances_fromqs");
             })
        .def("align", &FastGICP::align, py::arg("initial_guess") = py::none())
        .def("get_source_correspondence", &FastGICP::get_source_correspondence)
        .def(py::pickle(
            [](const FastGICP& g) { return py::cast(g.config); },                 // device state is per process: only the configuration travels
            [](py::object state) {
                auto cfg = state.cast<std::map<std::string, double>>();
                auto g = std::make_unique<FastGICP>();
                for (auto& kv : cfg) {
                    if (kv.first == "max_corr") g->set_max_correspondence_distance(kv.second);
                    else if (kv.first == "max_knn") g->set_max_knn_distance(kv.second);
                    else if (kv.first == "k") g->set_correspondence_randomness((int)kv.second);
                    else if (kv.first == "iters") g->set_max_iterations((int)kv.second);
                    else if (kv.first == "threads") g->set_num_threads((int)kv.second);
                }
                return g;
            }));
}
