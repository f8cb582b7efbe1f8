// ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (gs_icp_slam_amd/).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// CPU (C++17 + OpenMP, exact kd-tree) restatement of the GICP scan-to-model tracker behind `pygicp.FastGICP`,
// also used as the "fast_gicp OpenMP CPU path" stand-in for BASELINE.json config 1 (cpu_baseline.kind = "port").
//
// Pinned piece (golden vectors from the reference's own Python, tests/test_oracle_pinned.py): the (x,y,z,w) quaternion convention and
// R diag(s^2) R^T of set_target_covariances_fromqs against utils/general_utils.py:60-114.
// PARITY UNPINNED: /root/reference/submodules/fast_gicp is an EMPTY directory (fork
// Lab-of-AI-and-Robotics/fast_gicp, branch gs_icp_slam, commit unpinned: /root/reference/.gitmodules:4-7), its
// dependencies (PCL, Eigen, FLANN) are absent, and the reference has no tests.  This file follows
//   (i)  the call sites: mp_Tracker.py:53 (ctor), :109-110 (max_correspondence_distance, max_knn_distance),
//        :157-169 (set_input_target, set_target_filter, calculate_target_covariance_with_filter,
//        get_target_rotationsq/scales), :191-200 (set_input_source, set_source_filter, align), :231
//        (get_source_correspondence -> (indices, SQUARED distances), one per trackable source point),
//        :256-264 (get_source_rotationsq/scales, quaternions x,y,z,w), :287-288 (set_target_covariances_fromqs);
//   (ii) the published algorithm of SMRT-AIST/fast_gicp (FastGICP + LsqRegistration): k=20 nearest-neighbour
//        covariances, PLANE regularisation (singular values -> 1,1,1e-3), float-precision 1-NN correspondence with
//        gate d^2 < max_corr^2, Mahalanobis matrix (Sigma_B + T Sigma_A T^T)^-1, J = [skew(T a) | -I],
//        Levenberg-Marquardt with lambda0 = 1e-9 * max|diag H|, <= 10 inner trials, <= 64 outer iterations,
//        convergence max(|dR - I| / 2e-3, |dt| / 5e-4) < 1, final transform rounded through float.
// Fork-specific semantics that cannot be verified here: exported scales are sqrt(eigenvalues) of the
// RAW k-NN covariance (descending), quaternions (x,y,z,w) of the eigenvector frame with det = +1;
// set_target_covariances_fromqs builds R diag(s^2) R^T and applies the same regularisation as the k-NN path; target filter restricts which target points can be matched;
// max_knn_distance drops neighbours farther than that radius from the covariance estimate.
#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------ small linear algebra
struct M3 { double m[9]; };

inline M3 mul(const M3& a, const M3& b) {
    M3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
inline M3 transpose(const M3& a) {
    M3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * j + i];
    return c;
}
inline bool inv_sym3(const double* s /*xx xy xz yy yz zz*/, double* o) {
    const double a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5];
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double det = a * A + b * B + c * C;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = A * id; o[1] = B * id; o[2] = C * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
    return true;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (xx xy xz yy yz zz).  Eigenvalues descending, V columns =
// eigenvectors, det(V) = +1.  The HIP path runs the same fixed procedure so results agree to rounding.
inline void eig_sym3(const double* s, double* evals, double* V) {
    double a[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) {  // A <- A J
                const double akp = a[3 * k + p], akq = a[3 * k + q];
                a[3 * k + p] = c * akp - sn * akq;
                a[3 * k + q] = sn * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {  // A <- J^T A
                const double apk = a[3 * p + k], aqk = a[3 * q + k];
                a[3 * p + k] = c * apk - sn * aqk;
                a[3 * q + k] = sn * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[3 * k + p], vkq = v[3 * k + q];
                v[3 * k + p] = c * vkp - sn * vkq;
                v[3 * k + q] = sn * vkp + c * vkq;
            }
        }
    }
    int order[3] = {0, 1, 2};
    const double d[3] = {a[0], a[4], a[8]};
    std::sort(order, order + 3, [&](int i, int j) { return d[i] > d[j] || (d[i] == d[j] && i < j); });
    for (int c = 0; c < 3; ++c) {
        evals[c] = d[order[c]];
        for (int r = 0; r < 3; ++r) V[3 * r + c] = v[3 * r + order[c]];
    }
    const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (det < 0)
        for (int r = 0; r < 3; ++r) V[3 * r + 2] = -V[3 * r + 2];
}

inline void rot_to_quat_xyzw(const double* R, double* q) {
    const double t[4] = {1 + R[0] - R[4] - R[8], 1 - R[0] + R[4] - R[8], 1 - R[0] - R[4] + R[8], 1 + R[0] + R[4] + R[8]};
    int k = 0;
    for (int i = 1; i < 4; ++i)
        if (t[i] > t[k]) k = i;
    const double s = 2.0 * std::sqrt(t[k]);
    if (k == 0) { q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; q[3] = (R[7] - R[5]) / s; }
    else if (k == 1) { q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; q[3] = (R[2] - R[6]) / s; }
    else if (k == 2) { q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; q[3] = (R[3] - R[1]) / s; }
    else { q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; q[3] = 0.25 * s; }
}

inline void quat_xyzw_to_rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], r = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z); R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y); R[7] = 2 * (y * z + r * x); R[8] = 1 - 2 * (x * x + y * y);
}

// 6x6 LDL^T solve (no pivoting, as Eigen::LDLT on an SPD matrix up to its pivoting order)
inline bool solve6(const double* H, const double* b, double* x) {
    double L[36] = {0}, Dg[6];
    for (int j = 0; j < 6; ++j) {
        double d = H[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * Dg[k];
        Dg[j] = d;
        if (d == 0.0 || !std::isfinite(d)) return false;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double v = H[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * Dg[k];
            L[6 * i + j] = v / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; ++i) y[i] /= Dg[i];
    for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k]; x[i] = v; }
    return true;
}

struct Iso { double R[9]; double t[3]; };  // x -> R x + t

inline Iso iso_mul(const Iso& a, const Iso& b) {  // a * b
    Iso c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
        c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
    }
    return c;
}

inline Iso se3_exp(const double* a) {
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real, theta = 0;
    if (theta_sq < 1e-10) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = std::sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    const double q[4] = {imag * wx, imag * wy, imag * wz, real};
    Iso T;
    quat_xyzw_to_rot(q, T.R);
    double V[9];
    if (theta_sq < 1e-20) {  // theta < 1e-10
        std::memcpy(V, T.R, sizeof(V));
    } else {
        if (theta == 0) theta = std::sqrt(theta_sq);
        const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
        const double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) T.t[i] = V[3 * i] * a[3] + V[3 * i + 1] * a[4] + V[3 * i + 2] * a[5];
    return T;
}

// ------------------------------------------------------------------------------------------ exact kd-tree (float)
struct KdTree {
    struct Node { int lo, hi, left, right, dim; float split; };
    std::vector<Node> nodes;
    std::vector<int> idx;          // permutation of ORIGINAL indices
    const float* pts = nullptr;    // (n,3) original cloud
    static constexpr int LEAF = 10;

    void build(const float* p, const std::vector<int>& subset) {
        pts = p; idx = subset; nodes.clear();
        if (!idx.empty()) { nodes.reserve(2 * idx.size() / LEAF + 8); rec(0, (int)idx.size()); }
    }
    int rec(int lo, int hi) {
        const int id = (int)nodes.size();
        nodes.push_back(Node{lo, hi, -1, -1, 0, 0.f});
        if (hi - lo <= LEAF) return id;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = lo; i < hi; ++i)
            for (int d = 0; d < 3; ++d) { const float v = pts[3 * idx[i] + d]; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
        int dim = 0;
        for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
        const int mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                         [&](int a, int b) { return pts[3 * a + dim] < pts[3 * b + dim] || (pts[3 * a + dim] == pts[3 * b + dim] && a < b); });
        const float split = pts[3 * idx[mid] + dim];
        const int l = rec(lo, mid), r = rec(mid, hi);
        nodes[id].left = l; nodes[id].right = r; nodes[id].dim = dim; nodes[id].split = split;
        return id;
    }
    static inline float d2(const float* a, const float* b) {
        const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        return dx * dx + dy * dy + dz * dz;
    }
    // 1-NN, ties -> lowest original index
    void nn(const float* q, int node, float& best_d, int& best_i) const {
        const Node& n = nodes[node];
        if (n.left < 0) {
            for (int i = n.lo; i < n.hi; ++i) {
                const int id = idx[i];
                const float d = d2(q, pts + 3 * id);
                if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; }
            }
            return;
        }
        const float diff = q[n.dim] - n.split;
        const int first = diff < 0 ? n.left : n.right, second = diff < 0 ? n.right : n.left;
        nn(q, first, best_d, best_i);
        if (diff * diff <= best_d) nn(q, second, best_d, best_i);
    }
    // k-NN into a max-heap of (d2, index), lexicographic order
    void knn(const float* q, int node, int k, std::vector<std::pair<float, int>>& heap) const {
        const Node& n = nodes[node];
        if (n.left < 0) {
            for (int i = n.lo; i < n.hi; ++i) {
                const int id = idx[i];
                const std::pair<float, int> c(d2(q, pts + 3 * id), id);
                if ((int)heap.size() < k) { heap.push_back(c); std::push_heap(heap.begin(), heap.end()); }
                else if (c < heap.front()) { std::pop_heap(heap.begin(), heap.end()); heap.back() = c; std::push_heap(heap.begin(), heap.end()); }
            }
            return;
        }
        const float diff = q[n.dim] - n.split;
        const int first = diff < 0 ? n.left : n.right, second = diff < 0 ? n.right : n.left;
        knn(q, first, k, heap);
        if ((int)heap.size() < k || diff * diff <= heap.front().first) knn(q, second, k, heap);
    }
};

// ------------------------------------------------------------------------------------------ the registration object
struct Cloud {
    std::vector<float> pts;              // (n,3)
    int n = 0;
    std::vector<int> trackable;          // original indices of trackable points, in rank order
    std::vector<double> cov;             // (n,6) regularised covariance used by the cost
    std::vector<float> rotq, scales;     // (n,4) xyzw, (n,3)
    bool cov_valid = false;
    KdTree tree_all, tree_track;
    bool tree_all_valid = false, tree_track_valid = false;
};

struct Gicp {
    int k = 20, max_iter = 64, lm_max_iter = 10, reg = 3, threads = 0;
    int scale_mode = 0;           // 0: exported scales = sqrt(eigenvalues), fromqs squares them; 1: variances both ways (SURVEY 8a unknown)
    double max_corr = (double)FLT_MAX, max_knn = (double)FLT_MAX, rot_eps = 2e-3, trans_eps = 5e-4, lm_init = 1e-9;
    Cloud src, tgt;
    std::vector<int> corr;        // per trackable source point: target ORIGINAL index or -1
    std::vector<float> sqd;       // per trackable source point
    std::vector<double> maha;     // per trackable source point, 6
    double lm_lambda = -1;
    double H_final[36];
    double stats[6] = {0, 0, 0, 0, 0, 0};
};

void set_cloud(Cloud& c, const void* p, int n, int is_f64) {
    c.n = n; c.pts.resize((size_t)3 * n);
    if (is_f64) for (size_t i = 0; i < (size_t)3 * n; ++i) c.pts[i] = (float)((const double*)p)[i];
    else std::memcpy(c.pts.data(), p, sizeof(float) * 3 * n);
    c.trackable.resize(n);
    std::iota(c.trackable.begin(), c.trackable.end(), 0);
    c.cov_valid = false; c.tree_all_valid = c.tree_track_valid = false;
    c.cov.clear(); c.rotq.clear(); c.scales.clear();
}

void set_filter(Cloud& c, int n_track, const int32_t* f, int n) {
    c.trackable.assign(n_track, -1);
    for (int i = 0; i < n && i < c.n; ++i)
        if (f[i] > 0 && f[i] <= n_track) c.trackable[f[i] - 1] = i;
    c.trackable.erase(std::remove(c.trackable.begin(), c.trackable.end(), -1), c.trackable.end());
    c.tree_track_valid = false;
}

void ensure_trees(Cloud& c, bool all, bool track) {
    if (all && !c.tree_all_valid) {
        std::vector<int> ids(c.n);
        std::iota(ids.begin(), ids.end(), 0);
        c.tree_all.build(c.pts.data(), ids);
        c.tree_all_valid = true;
    }
    if (track && !c.tree_track_valid) { c.tree_track.build(c.pts.data(), c.trackable); c.tree_track_valid = true; }
}

void regularise(int method, const double* evals, const double* V, const double* raw, double* out6) {
    if (method == 0) { std::memcpy(out6, raw, 6 * sizeof(double)); return; }
    if (method == 4) {  // FROBENIUS: (C^-1 / ||C^-1||_F)^-1 with C = cov + 1e-3 I
        double C[6] = {raw[0] + 1e-3, raw[1], raw[2], raw[3] + 1e-3, raw[4], raw[5] + 1e-3}, Ci[6];
        inv_sym3(C, Ci);
        const double nrm = std::sqrt(Ci[0] * Ci[0] + Ci[3] * Ci[3] + Ci[5] * Ci[5] + 2 * (Ci[1] * Ci[1] + Ci[2] * Ci[2] + Ci[4] * Ci[4]));
        for (int i = 0; i < 6; ++i) Ci[i] /= nrm;
        inv_sym3(Ci, out6);
        return;
    }
    double vals[3];
    if (method == 3) { vals[0] = 1; vals[1] = 1; vals[2] = 1e-3; }
    else if (method == 1) { for (int i = 0; i < 3; ++i) vals[i] = std::max(evals[i], 1e-3); }
    else { const double mx = std::max(evals[0], 1e-300); for (int i = 0; i < 3; ++i) vals[i] = std::max(evals[i] / mx, 1e-3); }
    // V diag(vals) V^T
    int k = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) out6[k++] = V[3 * r] * vals[0] * V[3 * c] + V[3 * r + 1] * vals[1] * V[3 * c + 1] + V[3 * r + 2] * vals[2] * V[3 * c + 2];
}

void calc_cov(Gicp& g, Cloud& c) {
    ensure_trees(c, true, false);
    c.cov.assign((size_t)6 * c.n, 0); c.rotq.assign((size_t)4 * c.n, 0); c.scales.assign((size_t)3 * c.n, 0);
    const int k = std::min(g.k, c.n);
    const float maxd2 = g.max_knn >= (double)FLT_MAX ? FLT_MAX : (float)(g.max_knn * g.max_knn);
#pragma omp parallel
    {
        std::vector<std::pair<float, int>> heap;
#pragma omp for schedule(guided, 8)
        for (int i = 0; i < c.n; ++i) {
            heap.clear();
            c.tree_all.knn(c.pts.data() + 3 * i, 0, k, heap);
            std::sort(heap.begin(), heap.end());
            double mu[3] = {0, 0, 0};
            int cnt = 0;
            for (auto& h : heap) { if (h.first > maxd2) break; for (int d = 0; d < 3; ++d) mu[d] += c.pts[3 * h.second + d]; ++cnt; }
            for (int d = 0; d < 3; ++d) mu[d] /= cnt;
            double raw[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < cnt; ++j) {
                const float* p = c.pts.data() + 3 * heap[j].second;
                const double dx = p[0] - mu[0], dy = p[1] - mu[1], dz = p[2] - mu[2];
                raw[0] += dx * dx; raw[1] += dx * dy; raw[2] += dx * dz; raw[3] += dy * dy; raw[4] += dy * dz; raw[5] += dz * dz;
            }
            for (int d = 0; d < 6; ++d) raw[d] /= cnt;
            double ev[3], V[9], q[4];
            eig_sym3(raw, ev, V);
            rot_to_quat_xyzw(V, q);
            for (int d = 0; d < 4; ++d) c.rotq[4 * i + d] = (float)q[d];
            for (int d = 0; d < 3; ++d) c.scales[3 * i + d] = (float)(g.scale_mode ? std::max(ev[d], 0.0) : std::sqrt(std::max(ev[d], 0.0)));
            regularise(g.reg, ev, V, raw, c.cov.data() + 6 * i);
        }
    }
    c.cov_valid = true;
}

// correspondence search in float, Mahalanobis matrices in double
void update_corr(Gicp& g, const Iso& T) {
    ensure_trees(g.tgt, false, true);
    const int n = (int)g.src.trackable.size();
    g.corr.assign(n, -1); g.sqd.assign(n, FLT_MAX); g.maha.assign((size_t)6 * n, 0);
    float Rf[9], tf[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)T.R[i];
    for (int i = 0; i < 3; ++i) tf[i] = (float)T.t[i];
    const float gate = g.max_corr >= (double)FLT_MAX ? FLT_MAX : (float)g.max_corr * (float)g.max_corr;
    const bool empty = g.tgt.trackable.empty();
#pragma omp parallel for schedule(guided, 8)
    for (int s = 0; s < n; ++s) {
        const int i = g.src.trackable[s];
        const float* p = g.src.pts.data() + 3 * i;
        float q[3];
        for (int r = 0; r < 3; ++r) q[r] = ((Rf[3 * r] * p[0] + Rf[3 * r + 1] * p[1]) + Rf[3 * r + 2] * p[2]) + tf[r];
        float bd = FLT_MAX; int bi = -1;
        if (!empty) { bi = INT32_MAX; g.tgt.tree_track.nn(q, 0, bd, bi); }
        g.sqd[s] = bd;
        if (bi < 0 || bi == INT32_MAX || !(bd < gate)) continue;
        const double* A = g.src.cov.data() + 6 * i;
        const double* B = g.tgt.cov.data() + 6 * bi;
        const double Am[9] = {A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5]};
        double RA[9], RAR[6];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) RA[3 * r + c] = T.R[3 * r] * Am[c] + T.R[3 * r + 1] * Am[3 + c] + T.R[3 * r + 2] * Am[6 + c];
        int k = 0;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) RAR[k++] = RA[3 * r] * T.R[3 * c] + RA[3 * r + 1] * T.R[3 * c + 1] + RA[3 * r + 2] * T.R[3 * c + 2];
        double S[6];
        for (int d = 0; d < 6; ++d) S[d] = B[d] + RAR[d];
        if (!inv_sym3(S, g.maha.data() + 6 * s)) continue;
        g.corr[s] = bi;
    }
}

inline void residual(const Gicp& g, const Iso& T, int s, double* e, double* ta) {
    const int i = g.src.trackable[s];
    const float* a = g.src.pts.data() + 3 * i;
    const float* b = g.tgt.pts.data() + 3 * g.corr[s];
    for (int r = 0; r < 3; ++r) {
        ta[r] = T.R[3 * r] * (double)a[0] + T.R[3 * r + 1] * (double)a[1] + T.R[3 * r + 2] * (double)a[2] + T.t[r];
        e[r] = (double)b[r] - ta[r];
    }
}

double linearize(Gicp& g, const Iso& T, double* H, double* bvec) {
    update_corr(g, T);
    const int n = (int)g.src.trackable.size();
    double sum = 0;
    double Hs[36] = {0}, bs[6] = {0};
#pragma omp parallel
    {
        double Hl[36] = {0}, bl[6] = {0}, sl = 0;
#pragma omp for schedule(guided, 8) nowait
        for (int s = 0; s < n; ++s) {
            if (g.corr[s] < 0) continue;
            double e[3], ta[3];
            residual(g, T, s, e, ta);
            const double* m = g.maha.data() + 6 * s;
            const double Mm[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
            double Me[3];
            for (int r = 0; r < 3; ++r) Me[r] = Mm[3 * r] * e[0] + Mm[3 * r + 1] * e[1] + Mm[3 * r + 2] * e[2];
            sl += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
            // J (3x6) = [skew(ta) | -I]
            const double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
            double MJ[18];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 6; ++c) MJ[6 * r + c] = Mm[3 * r] * J[c] + Mm[3 * r + 1] * J[6 + c] + Mm[3 * r + 2] * J[12 + c];
            for (int r = 0; r < 6; ++r) {
                for (int c = 0; c < 6; ++c) Hl[6 * r + c] += J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c];
                bl[r] += J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2];
            }
        }
#pragma omp critical
        {
            for (int i = 0; i < 36; ++i) Hs[i] += Hl[i];
            for (int i = 0; i < 6; ++i) bs[i] += bl[i];
            sum += sl;
        }
    }
    std::memcpy(H, Hs, sizeof(Hs)); std::memcpy(bvec, bs, sizeof(bs));
    return sum;
}

double compute_error(const Gicp& g, const Iso& T) {
    const int n = (int)g.src.trackable.size();
    double sum = 0;
#pragma omp parallel for schedule(guided, 8) reduction(+ : sum)
    for (int s = 0; s < n; ++s) {
        if (g.corr[s] < 0) continue;
        double e[3], ta[3];
        residual(g, T, s, e, ta);
        const double* m = g.maha.data() + 6 * s;
        sum += e[0] * (m[0] * e[0] + m[1] * e[1] + m[2] * e[2]) + e[1] * (m[1] * e[0] + m[3] * e[1] + m[4] * e[2]) +
               e[2] * (m[2] * e[0] + m[4] * e[1] + m[5] * e[2]);
    }
    return sum;
}

bool is_converged(const Gicp& g, const Iso& d) {
    double mr = 0, mt = 0;
    for (int i = 0; i < 9; ++i) mr = std::max(mr, std::fabs(d.R[i] - (i % 4 == 0 ? 1.0 : 0.0)) / g.rot_eps);
    for (int i = 0; i < 3; ++i) mt = std::max(mt, std::fabs(d.t[i]) / g.trans_eps);
    return std::max(mr, mt) < 1.0;
}

bool step_lm(Gicp& g, Iso& x0, Iso& delta) {
    double H[36], b[6];
    const double y0 = linearize(g, x0, H, b);
    if (g.lm_lambda < 0.0) {
        double mx = 0;
        for (int i = 0; i < 6; ++i) mx = std::max(mx, std::fabs(H[7 * i]));
        g.lm_lambda = g.lm_init * mx;
    }
    double nu = 2.0;
    for (int it = 0; it < g.lm_max_iter; ++it) {
        g.stats[1] += 1;
        double Hl[36], nb[6], d[6];
        std::memcpy(Hl, H, sizeof(Hl));
        for (int i = 0; i < 6; ++i) { Hl[7 * i] += g.lm_lambda; nb[i] = -b[i]; }
        if (!solve6(Hl, nb, d)) return false;
        delta = se3_exp(d);
        const Iso xi = iso_mul(delta, x0);
        const double yi = compute_error(g, xi);
        double denom = 0;
        for (int i = 0; i < 6; ++i) denom += d[i] * (g.lm_lambda * d[i] - b[i]);
        const double rho = (y0 - yi) / denom;
        if (rho < 0) {
            if (is_converged(g, delta)) return true;
            g.lm_lambda = nu * g.lm_lambda;
            nu = 2 * nu;
            continue;
        }
        x0 = xi;
        g.lm_lambda = g.lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
        std::memcpy(g.H_final, H, sizeof(H));
        g.stats[2] = yi;
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

void* oracle_gicp_create() { return new Gicp(); }
void oracle_gicp_destroy(void* h) { delete (Gicp*)h; }
void oracle_gicp_set_param(void* h, int which, double v) {
    Gicp& g = *(Gicp*)h;
    switch (which) {
        case 0: g.max_corr = v; break;
        case 1: g.max_knn = v; g.src.cov_valid = g.tgt.cov_valid = false; break;
        case 2: g.k = (int)v; break;
        case 3: g.max_iter = (int)v; break;
        case 4: g.threads = (int)v; if (g.threads > 0) omp_set_num_threads(g.threads); break;
        case 5: g.reg = (int)v; break;
        case 6: g.rot_eps = v; break;
        case 7: g.trans_eps = v; break;
        case 8: g.scale_mode = (int)v; g.src.cov_valid = g.tgt.cov_valid = false; break;
    }
}
void oracle_gicp_set_input(void* h, int is_target, const void* p, int n, int is_f64) {
    Gicp& g = *(Gicp*)h;
    set_cloud(is_target ? g.tgt : g.src, p, n, is_f64);
}
void oracle_gicp_set_filter(void* h, int is_target, int n_track, const int32_t* f, int n) {
    Gicp& g = *(Gicp*)h;
    set_filter(is_target ? g.tgt : g.src, n_track, f, n);
}
void oracle_gicp_calc_cov(void* h, int is_target) {
    Gicp& g = *(Gicp*)h;
    calc_cov(g, is_target ? g.tgt : g.src);
}
int oracle_gicp_get_rotq(void* h, int is_target, float* out, int cap) {
    Cloud& c = is_target ? ((Gicp*)h)->tgt : ((Gicp*)h)->src;
    const int n = std::min(cap, (int)c.rotq.size() / 4);
    std::memcpy(out, c.rotq.data(), sizeof(float) * 4 * n);
    return n;
}
int oracle_gicp_get_scales(void* h, int is_target, float* out, int cap) {
    Cloud& c = is_target ? ((Gicp*)h)->tgt : ((Gicp*)h)->src;
    const int n = std::min(cap, (int)c.scales.size() / 3);
    std::memcpy(out, c.scales.data(), sizeof(float) * 3 * n);
    return n;
}
int oracle_gicp_get_cov(void* h, int is_target, double* out, int cap) {
    Cloud& c = is_target ? ((Gicp*)h)->tgt : ((Gicp*)h)->src;
    const int n = std::min(cap, (int)c.cov.size() / 6);
    std::memcpy(out, c.cov.data(), sizeof(double) * 6 * n);
    return n;
}
int oracle_gicp_set_target_cov_fromqs(void* h, const float* rots, int n_rots, const float* scales, int n_scales) {
    Gicp& g = *(Gicp*)h;
    const int K = g.tgt.n;
    if (n_rots != 4 * K || n_scales != 3 * K) return -1;
    g.tgt.cov.assign((size_t)6 * K, 0);
    g.tgt.rotq.assign(rots, rots + 4 * (size_t)K);
    g.tgt.scales.assign(scales, scales + 3 * (size_t)K);
    for (int i = 0; i < K; ++i) {
        double q[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
        const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (nrm > 0) for (int d = 0; d < 4; ++d) q[d] /= nrm; else { q[0] = q[1] = q[2] = 0; q[3] = 1; }
        double R[9];
        quat_xyzw_to_rot(q, R);
        const double s2[3] = {g.scale_mode ? (double)scales[3 * i] : (double)scales[3 * i] * scales[3 * i],
                              g.scale_mode ? (double)scales[3 * i + 1] : (double)scales[3 * i + 1] * scales[3 * i + 1],
                              g.scale_mode ? (double)scales[3 * i + 2] : (double)scales[3 * i + 2] * scales[3 * i + 2]};
        double raw[6];
        int k = 0;
        for (int r = 0; r < 3; ++r)
            for (int c = r; c < 3; ++c) raw[k++] = R[3 * r] * s2[0] * R[3 * c] + R[3 * r + 1] * s2[1] * R[3 * c + 1] + R[3 * r + 2] * s2[2] * R[3 * c + 2];
        // The same regularisation as the k-NN path, applied to the eigen-structure the Gaussian already carries:
        // eigenvalues = s^2 (stable descending order), eigenvectors = the matching columns of R.
        int o0 = 0, o1 = 1, o2 = 2;
        if (s2[o1] > s2[o0]) std::swap(o0, o1);
        if (s2[o2] > s2[o1]) std::swap(o1, o2);
        if (s2[o1] > s2[o0]) std::swap(o0, o1);
        const int order[3] = {o0, o1, o2};
        double ev[3], V[9];
        for (int c = 0; c < 3; ++c) { ev[c] = s2[order[c]]; for (int r = 0; r < 3; ++r) V[3 * r + c] = R[3 * r + order[c]]; }
        regularise(g.reg, ev, V, raw, g.tgt.cov.data() + 6 * i);
    }
    g.tgt.cov_valid = true;
    return 0;
}
// initial/final: row-major 4x4 double.  Returns outer iterations used.
int oracle_gicp_align(void* h, const double* init, double* out) {
    Gicp& g = *(Gicp*)h;
    if (!g.src.cov_valid) calc_cov(g, g.src);
    if (!g.tgt.cov_valid) calc_cov(g, g.tgt);
    Iso x0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) x0.R[3 * r + c] = (double)(float)init[4 * r + c];
        x0.t[r] = (double)(float)init[4 * r + 3];
    }
    g.lm_lambda = -1.0;
    g.stats[0] = g.stats[1] = g.stats[2] = g.stats[3] = 0;
    bool converged = false;
    int it = 0;
    for (; it < g.max_iter && !converged; ++it) {
        Iso delta;
        if (!step_lm(g, x0, delta)) break;
        converged = is_converged(g, delta);
    }
    g.stats[0] = it; g.stats[3] = converged ? 1 : 0;
    for (int i = 0; i < 16; ++i) out[i] = (i == 15) ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[4 * r + c] = (double)(float)x0.R[3 * r + c];
        out[4 * r + 3] = (double)(float)x0.t[r];
    }
    return it;
}
int oracle_gicp_get_corr(void* h, int32_t* idx, float* d2, int cap) {
    Gicp& g = *(Gicp*)h;
    const int n = std::min(cap, (int)g.corr.size());
    std::memcpy(idx, g.corr.data(), sizeof(int32_t) * n);
    std::memcpy(d2, g.sqd.data(), sizeof(float) * n);
    return n;
}
void oracle_gicp_stats(void* h, double* out) { std::memcpy(out, ((Gicp*)h)->stats, sizeof(double) * 6); }
void oracle_gicp_hessian(void* h, double* out) { std::memcpy(out, ((Gicp*)h)->H_final, sizeof(double) * 36); }
int oracle_gicp_num_threads() { return omp_get_max_threads(); }

}  // extern "C"
