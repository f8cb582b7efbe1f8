// GICP scan-to-model tracker on gfx950 — replaces pygicp.FastGICP [REF mp_Tracker.py:53] (the reference runs
// fast_gicp's CPU/OpenMP kd-tree path [REF docker_folder/Dockerfile:41-49]; there is no GPU tracker to port).
//
// Call sites mirrored: set_max_correspondence_distance / set_max_knn_distance [REF mp_Tracker.py:109-110],
// set_input_target + set_target_filter + calculate_target_covariance_with_filter + get_target_rotationsq/scales
// [REF mp_Tracker.py:157-169], set_input_source + set_source_filter + align [REF mp_Tracker.py:191-200],
// get_source_correspondence [REF mp_Tracker.py:231], get_source_rotationsq/scales [REF mp_Tracker.py:256-264],
// set_target_covariances_fromqs [REF mp_Tracker.py:287-288].
//
// MI355X design (DESIGN.md has the numbers):
//  * Neighbour search is an exact, radius-bounded 1-NN on a hashed uniform grid (cell = 2 x gate radius, 8-cell
//    probe chosen by which half of the cell the query falls in), not a kd-tree: the problem is 8-12 k queries
//    against <= 1 M targets, which is latency- not bandwidth-bound, and a hash probe is a handful of dependent
//    L2 hits where a kd-tree descent is ~20.  Points are stored sorted by cell so candidates are contiguous.
//  * One align() = ONE kernel launch: a persistent 1024-thread workgroup runs every Levenberg-Marquardt iteration
//    on the device (correspondences, Mahalanobis matrices, 6x6 normal equations by wave-shuffle + LDS tree
//    reduction in fp64, LDL^T solve, SE(3) exponential, trial-cost evaluation, accept/reject), so a frame costs
//    one launch and one 512-byte read-back instead of ~64 x 11 launches and host round-trips.
//  * k-NN covariances (k = 20 over <= 12 k points) are an LDS-tiled brute force with a register-resident sorted
//    top-k per thread; the 3x3 eigen-decomposition (cyclic Jacobi, fp64) is fused into the same kernel.
//  * Per-point arithmetic follows the reference algorithm's precision: correspondence search in fp32 with a fixed
//    evaluation order (this file is compiled with -ffp-contract=off, so nearest-neighbour indices are bit-exact
//    against the CPU oracle), cost / Jacobians / reductions in fp64.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/gsicp_hip.h"
#include "raster_common.hpp"

namespace gsicp {
extern thread_local std::string g_last_error;
}
using gsicp::g_last_error;

#define GC(expr)                                                                                                  \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                   \
            g_last_error = std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (gicp.hip:" +             \
                           std::to_string(__LINE__) + ")";                                                        \
            return -1;                                                                                            \
        }                                                                                                         \
    } while (0)

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr int NRED = 28;  // 21 (upper H) + 6 (b) + 1 (cost)

struct GridView {
    int use_grid;             // 0 -> brute force over `n_sorted` points
    float inv_h;              // 1 / cell size
    unsigned mask;            // table capacity - 1
    const unsigned long long* keys;
    const unsigned long long* vals;   // start << 32 | count
    const float4* sorted;     // xyz + original index bits, sorted by cell key
    int n_sorted;
};

// ---------------------------------------------------------------------------------------------- device helpers
__device__ inline unsigned long long cell_key(int cx, int cy, int cz) {
    const unsigned long long o = 1ull << 20;
    return ((unsigned long long)(cx + (long long)o) & 0x1FFFFFull) << 42 | ((unsigned long long)(cy + (long long)o) & 0x1FFFFFull) << 21 |
           ((unsigned long long)(cz + (long long)o) & 0x1FFFFFull);
}
__device__ inline unsigned hash_key(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}
__device__ inline float dist2(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return dx * dx + dy * dy + dz * dz;
}

// Exact nearest neighbour among the 8 cells around q (complete for every target point within h/2 of q).
// Ties -> lowest original index.  Returns best squared distance / index (FLT_MAX / -1 if the cells are empty).
__device__ inline void grid_nn(const GridView& g, float qx, float qy, float qz, float& best_d, int& best_i) {
    best_d = FLT_MAX; best_i = -1;
    if (!g.use_grid) {
        for (int j = 0; j < g.n_sorted; ++j) {
            const float4 p = g.sorted[j];
            const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
            const int id = __float_as_int(p.w);
            if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; }
        }
        return;
    }
    const float fx = qx * g.inv_h, fy = qy * g.inv_h, fz = qz * g.inv_h;
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
    const int ox = (fx - flx) >= 0.5f ? 1 : -1, oy = (fy - fly) >= 0.5f ? 1 : -1, oz = (fz - flz) >= 0.5f ? 1 : -1;
    unsigned long long cell[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {   // issue all 8 first probes together (independent loads)
        const unsigned long long key = cell_key(cx + ((c & 1) ? ox : 0), cy + ((c & 2) ? oy : 0), cz + ((c & 4) ? oz : 0));
        unsigned slot = hash_key(key) & g.mask;
        unsigned long long v = 0ull;
        for (;;) {
            const unsigned long long k = g.keys[slot];
            if (k == key) { v = g.vals[slot]; break; }
            if (k == EMPTY_KEY) break;
            slot = (slot + 1) & g.mask;
        }
        cell[c] = v;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned start = (unsigned)(cell[c] >> 32), cnt = (unsigned)cell[c];
        for (unsigned j = 0; j < cnt; ++j) {
            const float4 p = g.sorted[start + j];
            const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
            const int id = __float_as_int(p.w);
            if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; }
        }
    }
}

__device__ inline bool inv_sym3(const double* s, double* o) {
    const double a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5];
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double det = a * A + b * B + c * C;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = A * id; o[1] = B * id; o[2] = C * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
    return true;
}

__device__ inline void quat_to_rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], r = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z); R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y); R[7] = 2 * (y * z + r * x); R[8] = 1 - 2 * (x * x + y * y);
}

__device__ inline void rot_to_quat(const double* R, double* q) {
    const double t[4] = {1 + R[0] - R[4] - R[8], 1 - R[0] + R[4] - R[8], 1 - R[0] - R[4] + R[8], 1 + R[0] + R[4] + R[8]};
    int k = 0;
    for (int i = 1; i < 4; ++i)
        if (t[i] > t[k]) k = i;
    const double s = 2.0 * sqrt(t[k]);
    if (k == 0) { q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; q[3] = (R[7] - R[5]) / s; }
    else if (k == 1) { q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; q[3] = (R[2] - R[6]) / s; }
    else if (k == 2) { q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; q[3] = (R[3] - R[1]) / s; }
    else { q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; q[3] = 0.25 * s; }
}

// Cyclic Jacobi, identical procedure to the oracle's eig_sym3 (eigenvalues descending, det(V) = +1).
__device__ inline void eig_sym3(const double* s, double* evals, double* V) {
    double a[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double akp = a[3 * k + p], akq = a[3 * k + q];
                a[3 * k + p] = c * akp - sn * akq;
                a[3 * k + q] = sn * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double apk = a[3 * p + k], aqk = a[3 * q + k];
                a[3 * p + k] = c * apk - sn * aqk;
                a[3 * q + k] = sn * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[3 * k + p], vkq = v[3 * k + q];
                v[3 * k + p] = c * vkp - sn * vkq;
                v[3 * k + q] = sn * vkp + c * vkq;
            }
        }
    }
    const double d[3] = {a[0], a[4], a[8]};
    // stable descending order of three values
    int o0 = 0, o1 = 1, o2 = 2;
    if (d[o1] > d[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (d[o2] > d[o1]) { const int t = o1; o1 = o2; o2 = t; }
    if (d[o1] > d[o0]) { const int t = o0; o0 = o1; o1 = t; }
    const int order[3] = {o0, o1, o2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        evals[c] = d[order[c]];
#pragma unroll
        for (int r = 0; r < 3; ++r) V[3 * r + c] = v[3 * r + order[c]];
    }
    const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    if (det < 0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }
}

__device__ inline void regularise(int method, const double* evals, const double* V, const double* raw, double* out6) {
    if (method == 0) { for (int i = 0; i < 6; ++i) out6[i] = raw[i]; return; }
    if (method == 4) {
        double C[6] = {raw[0] + 1e-3, raw[1], raw[2], raw[3] + 1e-3, raw[4], raw[5] + 1e-3}, Ci[6];
        inv_sym3(C, Ci);
        const double nrm = sqrt(Ci[0] * Ci[0] + Ci[3] * Ci[3] + Ci[5] * Ci[5] + 2 * (Ci[1] * Ci[1] + Ci[2] * Ci[2] + Ci[4] * Ci[4]));
        for (int i = 0; i < 6; ++i) Ci[i] /= nrm;
        inv_sym3(Ci, out6);
        return;
    }
    double vals[3];
    if (method == 3) { vals[0] = 1; vals[1] = 1; vals[2] = 1e-3; }
    else if (method == 1) { for (int i = 0; i < 3; ++i) vals[i] = fmax(evals[i], 1e-3); }
    else { const double mx = fmax(evals[0], 1e-300); for (int i = 0; i < 3; ++i) vals[i] = fmax(evals[i] / mx, 1e-3); }
    int k = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = r; c < 3; ++c)
            out6[k++] = V[3 * r] * vals[0] * V[3 * c] + V[3 * r + 1] * vals[1] * V[3 * c + 1] + V[3 * r + 2] * vals[2] * V[3 * c + 2];
}

// ---------------------------------------------------------------------------------------------- k-NN covariances
// One WAVE per query point (4 queries per 256-thread workgroup, so ~n/4 workgroups fill the chip even at n = 8 k).
// The 64 lanes stream the cloud 64 candidates at a time (one coalesced 1 KiB load); the running top-k is a sorted
// list held ACROSS the lanes (lane r = rank r), so an insertion is a ballot + popcount + one wave_shr DPP move
// instead of a 20-deep per-thread compare chain.  Ordering is lexicographic (d2, index) like the oracle's.
__device__ inline bool lex_less(float d, int i, float d2, int i2) { return d < d2 || (d == d2 && i < i2); }
__device__ inline float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }  // lane l <- lane l-1

__global__ __launch_bounds__(256) void knn_cov_kernel(int n, int k, int batch_stride, const float4* __restrict__ pts,
                                                      int* __restrict__ nbr_idx, float* __restrict__ nbr_d2) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;                       // wave-uniform
    const int kk = k < n ? k : n;             // <= 64
    const unsigned long long kmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
    const float4 Q = pts[q];
    float my_d = FLT_MAX;
    int my_i = 0x7fffffff;
    float tau_d = FLT_MAX;
    int tau_i = 0x7fffffff;
    // Batches are visited in a multiplicative-permutation order: depth-image clouds arrive in raster order, and a
    // monotone approach towards the query would make almost every candidate an insertion.  The final sorted list
    // does not depend on the visiting order (the order is a total one).
    const int nb = (n + 63) / 64;
    // software pipeline: the load of the next batch is in flight while this one is tested / inserted
    float4 pnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n) pnext = pts[lane];
    for (int b = 0, pb = 0; b < nb; ++b) {
        const int j = pb * 64 + lane;
        const float4 p = pnext;
        pb += batch_stride;
        if (pb >= nb) pb -= nb;
        if (b + 1 < nb && pb * 64 + lane < n) pnext = pts[pb * 64 + lane];
        float d = FLT_MAX;
        int id = 0x7fffffff;
        if (j < n) {
            d = dist2(Q.x, Q.y, Q.z, p.x, p.y, p.z);
            id = j;
        }
        unsigned long long m = __ballot(lex_less(d, id, tau_d, tau_i));
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float cd = readlane_f(d, b);
            const int ci = __builtin_amdgcn_readlane(id, b);
            if (!lex_less(cd, ci, tau_d, tau_i)) continue;   // the list moved on since the ballot
            const unsigned long long le = __ballot(!lex_less(cd, ci, my_d, my_i)) & kmask;   // entries ranked before the candidate
            const int pos = __popcll(le);
            const float up_d = __int_as_float(wave_shr1(__float_as_int(my_d)));
            const int up_i = wave_shr1(my_i);
            if (lane > pos) { my_d = up_d; my_i = up_i; }
            else if (lane == pos) { my_d = cd; my_i = ci; }
            tau_d = readlane_f(my_d, kk - 1);
            tau_i = __builtin_amdgcn_readlane(my_i, kk - 1);
        }
    }
    // neighbours now sit in lanes 0..kk-1, ascending: park (index, d2) for the per-thread covariance kernel.  (Doing the fp64
    // mean / covariance / Jacobi here would execute it once per WAVE with all 64 lanes computing the same numbers.)
    if (lane < kk) {
        nbr_idx[(size_t)q * 64 + lane] = my_i;
        nbr_d2[(size_t)q * 64 + lane] = my_d;
    }
}

// One THREAD per point: mean / covariance of its neighbours in fp64 (summed in rank order, as the oracle does), cyclic
// Jacobi, quaternion (x,y,z,w), scales = sqrt(eigenvalues of the RAW covariance), regularised covariance for the cost.
__global__ __launch_bounds__(64) void cov_eig_kernel(int n, int k, const float4* __restrict__ pts, const int* __restrict__ nbr_idx,
                                                     const float* __restrict__ nbr_d2, float max_d2, int reg_method,
                                                     double* __restrict__ cov, float* __restrict__ rotq, float* __restrict__ scales) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= n) return;
    const int kk = k < n ? k : n;
    double mu[3] = {0, 0, 0};
    int cnt = 0;
    for (int j = 0; j < kk; ++j) {
        if (nbr_d2[(size_t)q * 64 + j] > max_d2) break;
        const float4 p = pts[nbr_idx[(size_t)q * 64 + j]];
        mu[0] += (double)p.x; mu[1] += (double)p.y; mu[2] += (double)p.z;
        ++cnt;
    }
    mu[0] /= cnt; mu[1] /= cnt; mu[2] /= cnt;
    double raw[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < cnt; ++j) {
        const float4 p = pts[nbr_idx[(size_t)q * 64 + j]];
        const double dx = (double)p.x - mu[0], dy = (double)p.y - mu[1], dz = (double)p.z - mu[2];
        raw[0] += dx * dx; raw[1] += dx * dy; raw[2] += dx * dz; raw[3] += dy * dy; raw[4] += dy * dz; raw[5] += dz * dz;
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) raw[d] /= cnt;
    double ev[3], V[9], qd[4], out6[6];
    eig_sym3(raw, ev, V);
    rot_to_quat(V, qd);
    regularise(reg_method, ev, V, raw, out6);
#pragma unroll
    for (int d = 0; d < 4; ++d) rotq[4 * (size_t)q + d] = (float)qd[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) scales[3 * (size_t)q + d] = (float)sqrt(fmax(ev[d], 0.0));
#pragma unroll
    for (int d = 0; d < 6; ++d) cov[6 * (size_t)q + d] = out6[d];
}

__global__ __launch_bounds__(256) void cov_fromqs_kernel(int n, int reg_method, const float* __restrict__ rots,
                                                         const float* __restrict__ scales, double* __restrict__ cov) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double q[4] = {rots[4 * (size_t)i], rots[4 * (size_t)i + 1], rots[4 * (size_t)i + 2], rots[4 * (size_t)i + 3]};
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nrm > 0) { q[0] /= nrm; q[1] /= nrm; q[2] /= nrm; q[3] /= nrm; } else { q[0] = q[1] = q[2] = 0; q[3] = 1; }
    double R[9];
    quat_to_rot(q, R);
    const double s0 = scales[3 * (size_t)i], s1 = scales[3 * (size_t)i + 1], s2 = scales[3 * (size_t)i + 2];
    const double v[3] = {s0 * s0, s1 * s1, s2 * s2};
    double raw[6];
    int k = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = r; c < 3; ++c)
            raw[k++] = R[3 * r] * v[0] * R[3 * c] + R[3 * r + 1] * v[1] * R[3 * c + 1] + R[3 * r + 2] * v[2] * R[3 * c + 2];
    // same regularisation as the k-NN path, on the eigen-structure the Gaussian carries (s^2 descending, columns of R)
    int o0 = 0, o1 = 1, o2 = 2;
    if (v[o1] > v[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (v[o2] > v[o1]) { const int t = o1; o1 = o2; o2 = t; }
    if (v[o1] > v[o0]) { const int t = o0; o0 = o1; o1 = t; }
    const int order[3] = {o0, o1, o2};
    double ev[3], V[9], out6[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ev[c] = v[order[c]];
#pragma unroll
        for (int r = 0; r < 3; ++r) V[3 * r + c] = R[3 * r + order[c]];
    }
    regularise(reg_method, ev, V, raw, out6);
#pragma unroll
    for (int d = 0; d < 6; ++d) cov[6 * (size_t)i + d] = out6[d];
}

// ---------------------------------------------------------------------------------------------- hash grid build
__global__ __launch_bounds__(256) void grid_keys_kernel(int n_track, const int* __restrict__ track, const float4* __restrict__ pts,
                                                        float inv_h, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_track) return;
    const int i = track[s];
    const float4 p = pts[i];
    keys[s] = cell_key((int)floorf(p.x * inv_h), (int)floorf(p.y * inv_h), (int)floorf(p.z * inv_h));
    vals[s] = (unsigned)i;
}
// sorted (key, original index) -> sorted float4 records; first element of each run inserts {start,count} in the table
__global__ __launch_bounds__(256) void grid_fill_kernel(int n, const unsigned long long* __restrict__ skeys, const unsigned* __restrict__ svals,
                                                        const float4* __restrict__ pts, float4* __restrict__ sorted, unsigned mask,
                                                        unsigned long long* __restrict__ tkeys, unsigned long long* __restrict__ tvals) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const unsigned id = svals[s];
    float4 p = pts[id];
    p.w = __int_as_float((int)id);
    sorted[s] = p;
    const unsigned long long key = skeys[s];
    if (s == 0 || skeys[s - 1] != key) {
        unsigned cnt = 1;
        while (s + (int)cnt < n && skeys[s + cnt] == key) ++cnt;
        unsigned slot = hash_key(key) & mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&tkeys[slot], EMPTY_KEY, key);
            if (prev == EMPTY_KEY) break;
            slot = (slot + 1) & mask;
        }
        tvals[slot] = ((unsigned long long)(unsigned)s << 32) | cnt;
    }
}
// brute-force "grid": just the trackable points in index order
__global__ __launch_bounds__(256) void gather_track_kernel(int n_track, const int* __restrict__ track, const float4* __restrict__ pts,
                                                           float4* __restrict__ sorted) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_track) return;
    const int i = track[s];
    float4 p = pts[i];
    p.w = __int_as_float(i);
    sorted[s] = p;
}

// ---------------------------------------------------------------------------------------------- align (persistent)
struct AlignResult {
    double final_pose[16];     // row-major 4x4, rounded through float
    double lin_pose[12];       // R (9) + t (3) of the last linearisation (what correspondences / distances refer to)
    double H_final[36];
    double cost;
    int iterations, lm_trials, converged, failed;
};

constexpr int AL_T = 256;        // threads per workgroup
constexpr int AL_MAX_WG = 240;   // <= one workgroup per CU, so every workgroup is resident and the grid barrier is safe

struct AlignSync {               // zeroed by a hipMemsetAsync before every launch
    unsigned counter;
    unsigned abort;
    unsigned pad[30];
    double partials[2][AL_MAX_WG][NRED];
};

struct AlignArgs {
    int n_src;                 // trackable source points
    const int* src_track;
    const float4* src_pts;
    const double* src_cov;
    const float4* tgt_pts;     // original order
    const double* tgt_cov;
    GridView grid;
    float gate;                // squared correspondence gate (FLT_MAX = none)
    double init[12];           // R, t
    int max_iter, lm_max_iter;
    double rot_eps, trans_eps, lm_init;
    int* corr;                 // per trackable source point
    float* sqd;
    double* maha;              // 6 per trackable source point
    AlignResult* result;
    AlignSync* sync;
};

__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ inline bool solve6(const double* H, const double* b, double* x) {
    double L[36], Dg[6];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int j = 0; j < 6; ++j) {
        double d = H[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * Dg[k];
        Dg[j] = d;
        if (d == 0.0 || !isfinite(d)) return false;
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

__device__ inline void se3_exp(const double* a, double* R, double* t) {
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real, theta = 0;
    if (theta_sq < 1e-10) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    const double q[4] = {imag * wx, imag * wy, imag * wz, real};
    quat_to_rot(q, R);
    double V[9];
    if (theta_sq < 1e-20) {
        for (int i = 0; i < 9; ++i) V[i] = R[i];
    } else {
        if (theta == 0) theta = sqrt(theta_sq);
        const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
        const double c1 = (1.0 - cos(theta)) / theta_sq, c2 = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) t[i] = V[3 * i] * a[3] + V[3 * i + 1] * a[4] + V[3 * i + 2] * a[5];
}

__device__ inline bool is_converged(const double* R, const double* t, double rot_eps, double trans_eps) {
    double mr = 0, mt = 0;
    for (int i = 0; i < 9; ++i) mr = fmax(mr, fabs(R[i] - (i % 4 == 0 ? 1.0 : 0.0)) / rot_eps);
    for (int i = 0; i < 3; ++i) mt = fmax(mt, fabs(t[i]) / trans_eps);
    return fmax(mr, mt) < 1.0;
}

struct AlignShared {
    double scratch[AL_T / 64][NRED];
    double red[NRED];
    double x0[12];      // current pose R,t
    double xi[12];      // trial pose
    double delta[12];
    double H[36], b[6];
    double y0, lambda, nu, denom;
    int state;          // 0 continue LM trials, 1 step accepted / done with this outer iteration, 2 abort
    int converged;
    int abort;
};

// Grid-wide sum of NV doubles per thread.  Workgroup partials go to global memory, one monotonic-counter grid barrier
// (agent-scope release before the arrive, relaxed polling by ONE lane, agent-scope acquire after — the protocol of
// cdna_hip_programming.md §6 G16), then every workgroup adds the partials in the same fixed order, so all workgroups
// hold bit-identical totals and take identical decisions without any further communication.
template <int NV>
__device__ inline void grid_sum(double* vals, AlignShared& sh, AlignSync* sy, unsigned& epoch, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum_d(vals[k]);
        if (lane == 0) sh.scratch[wave][k] = s;
    }
    __syncthreads();
    const int buf = epoch & 1;
    if (tid < NV) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < AL_T / 64; ++w) s += sh.scratch[w][tid];
        sy->partials[buf][blockIdx.x][tid] = s;
    }
    ++epoch;
    if (nwg > 1) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&sy->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * (unsigned)nwg;
            unsigned spins = 0;
            int ab = 0;
            while (__hip_atomic_load(&sy->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000000u || __hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ab = 1;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sh.abort = ab;
        }
    }
    __syncthreads();
    if (tid < NV) {
        double s = 0;
        for (int w = 0; w < nwg; ++w) s += sy->partials[buf][w][tid];
        sh.red[tid] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(AL_T) void gicp_align_kernel(AlignArgs a) {
    __shared__ AlignShared sh;
    const int tid = threadIdx.x;
    const int gtid = blockIdx.x * AL_T + tid, gstride = gridDim.x * AL_T;
    const bool leader = blockIdx.x == 0 && tid == 0;
    unsigned epoch = 0;
    if (tid < 12) sh.x0[tid] = a.init[tid];
    if (tid == 0) { sh.lambda = -1.0; sh.converged = 0; sh.abort = 0; }
    __syncthreads();

    int iterations = 0, lm_trials = 0, failed = 0;

    for (int it = 0; it < a.max_iter; ++it) {
        // ---------------- linearize at x0: correspondences + Mahalanobis + H, b, cost
        double R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = sh.x0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = sh.x0[9 + i];
        float Rf[9], tf[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tf[i] = (float)t[i];
        double acc[NRED];
#pragma unroll
        for (int k = 0; k < NRED; ++k) acc[k] = 0;

        for (int s = gtid; s < a.n_src; s += gstride) {
            const int i = a.src_track[s];
            const float4 p = a.src_pts[i];
            const float qx = ((Rf[0] * p.x + Rf[1] * p.y) + Rf[2] * p.z) + tf[0];
            const float qy = ((Rf[3] * p.x + Rf[4] * p.y) + Rf[5] * p.z) + tf[1];
            const float qz = ((Rf[6] * p.x + Rf[7] * p.y) + Rf[8] * p.z) + tf[2];
            float bd; int bi;
            grid_nn(a.grid, qx, qy, qz, bd, bi);
            a.sqd[s] = bd;
            int c = -1;
            if (bi >= 0 && bd < a.gate) {
                const double* A = a.src_cov + 6 * (size_t)i;
                const double* B = a.tgt_cov + 6 * (size_t)bi;
                const double Am[9] = {A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5]};
                double RA[9], S[6];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) RA[3 * r + cc] = R[3 * r] * Am[cc] + R[3 * r + 1] * Am[3 + cc] + R[3 * r + 2] * Am[6 + cc];
                int k = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int cc = r; cc < 3; ++cc) {
                        S[k] = B[k] + (RA[3 * r] * R[3 * cc] + RA[3 * r + 1] * R[3 * cc + 1] + RA[3 * r + 2] * R[3 * cc + 2]);
                        ++k;
                    }
                double m[6];
                if (inv_sym3(S, m)) {
                    c = bi;
#pragma unroll
                    for (int d = 0; d < 6; ++d) a.maha[6 * (size_t)s + d] = m[d];
                    const float4 bp = a.tgt_pts[bi];
                    double ta[3], e[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) ta[r] = R[3 * r] * (double)p.x + R[3 * r + 1] * (double)p.y + R[3 * r + 2] * (double)p.z + t[r];
                    e[0] = (double)bp.x - ta[0]; e[1] = (double)bp.y - ta[1]; e[2] = (double)bp.z - ta[2];
                    const double Mm[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
                    double Me[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) Me[r] = Mm[3 * r] * e[0] + Mm[3 * r + 1] * e[1] + Mm[3 * r + 2] * e[2];
                    acc[27] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
                    const double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
                    double MJ[18];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int cc = 0; cc < 6; ++cc) MJ[6 * r + cc] = Mm[3 * r] * J[cc] + Mm[3 * r + 1] * J[6 + cc] + Mm[3 * r + 2] * J[12 + cc];
                    int kk = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
#pragma unroll
                        for (int cc = r; cc < 6; ++cc) { acc[kk] += J[r] * MJ[cc] + J[6 + r] * MJ[6 + cc] + J[12 + r] * MJ[12 + cc]; ++kk; }
                        acc[21 + r] += J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2];
                    }
                }
            }
            a.corr[s] = c;
        }
        grid_sum<NRED>(acc, sh, a.sync, epoch, tid);
        if (sh.abort) { failed = 2; break; }
        if (tid == 0) {
            int kk = 0;
            for (int r = 0; r < 6; ++r)
                for (int cc = r; cc < 6; ++cc) { sh.H[6 * r + cc] = sh.red[kk]; sh.H[6 * cc + r] = sh.red[kk]; ++kk; }
            for (int r = 0; r < 6; ++r) sh.b[r] = sh.red[21 + r];
            sh.y0 = sh.red[27];
            if (sh.lambda < 0.0) {
                double mx = 0;
                for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(sh.H[7 * i]));
                sh.lambda = a.lm_init * mx;
            }
            sh.nu = 2.0;
            if (leader) for (int i = 0; i < 12; ++i) a.result->lin_pose[i] = sh.x0[i];
        }
        __syncthreads();

        // ---------------- LM trials (every workgroup runs the same scalar arithmetic on the same totals)
        bool step_ok = false;
        for (int trial = 0; trial < a.lm_max_iter; ++trial) {
            ++lm_trials;
            if (tid == 0) {
                double Hl[36], nb[6], d[6];
                for (int i = 0; i < 36; ++i) Hl[i] = sh.H[i];
                for (int i = 0; i < 6; ++i) { Hl[7 * i] += sh.lambda; nb[i] = -sh.b[i]; }
                if (!solve6(Hl, nb, d)) {
                    sh.state = 2;
                } else {
                    sh.state = 0;
                    se3_exp(d, sh.delta, sh.delta + 9);
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j)
                            sh.xi[3 * i + j] = sh.delta[3 * i] * sh.x0[j] + sh.delta[3 * i + 1] * sh.x0[3 + j] + sh.delta[3 * i + 2] * sh.x0[6 + j];
                        sh.xi[9 + i] = sh.delta[3 * i] * sh.x0[9] + sh.delta[3 * i + 1] * sh.x0[10] + sh.delta[3 * i + 2] * sh.x0[11] + sh.delta[9 + i];
                    }
                    double den = 0;
                    for (int i = 0; i < 6; ++i) den += d[i] * (sh.lambda * d[i] - sh.b[i]);
                    sh.denom = den;
                }
            }
            __syncthreads();
            if (sh.state == 2) break;
            // trial cost with frozen correspondences / Mahalanobis matrices
            double Rx[9], tx[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rx[i] = sh.xi[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) tx[i] = sh.xi[9 + i];
            double cost[1] = {0};
            for (int s = gtid; s < a.n_src; s += gstride) {
                const int c = a.corr[s];
                if (c < 0) continue;
                const float4 p = a.src_pts[a.src_track[s]];
                const float4 bp = a.tgt_pts[c];
                double e[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) e[r] = -(Rx[3 * r] * (double)p.x + Rx[3 * r + 1] * (double)p.y + Rx[3 * r + 2] * (double)p.z + tx[r]);
                e[0] += (double)bp.x; e[1] += (double)bp.y; e[2] += (double)bp.z;
                const double* m = a.maha + 6 * (size_t)s;
                cost[0] += e[0] * (m[0] * e[0] + m[1] * e[1] + m[2] * e[2]) + e[1] * (m[1] * e[0] + m[3] * e[1] + m[4] * e[2]) +
                           e[2] * (m[2] * e[0] + m[4] * e[1] + m[5] * e[2]);
            }
            grid_sum<1>(cost, sh, a.sync, epoch, tid);
            if (sh.abort) { failed = 2; break; }
            if (tid == 0) {
                const double yi = sh.red[0];
                const double rho = (sh.y0 - yi) / sh.denom;
                if (rho < 0) {
                    if (is_converged(sh.delta, sh.delta + 9, a.rot_eps, a.trans_eps)) {
                        sh.state = 1;       // upstream returns true without accepting the step
                    } else {
                        sh.lambda = sh.nu * sh.lambda;
                        sh.nu = 2 * sh.nu;
                        sh.state = 0;
                    }
                } else {
                    for (int i = 0; i < 12; ++i) sh.x0[i] = sh.xi[i];
                    const double f = 2 * rho - 1;
                    sh.lambda = sh.lambda * fmax(1.0 / 3.0, 1 - f * f * f);
                    if (leader) {
                        for (int i = 0; i < 36; ++i) a.result->H_final[i] = sh.H[i];
                        a.result->cost = yi;
                    }
                    sh.state = 1;
                }
            }
            __syncthreads();
            if (sh.state == 1) { step_ok = true; break; }
        }
        if (failed) break;
        if (!step_ok) { failed = 1; break; }   // "lm not converged"
        ++iterations;
        if (tid == 0) sh.converged = is_converged(sh.delta, sh.delta + 9, a.rot_eps, a.trans_eps) ? 1 : 0;
        __syncthreads();
        if (sh.converged) break;
    }
    if (leader) {
        AlignResult* r = a.result;
        for (int i = 0; i < 16; ++i) r->final_pose[i] = (i == 15) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r->final_pose[4 * i + j] = (double)(float)sh.x0[3 * i + j];
            r->final_pose[4 * i + 3] = (double)(float)sh.x0[9 + i];
        }
        r->iterations = iterations; r->lm_trials = lm_trials; r->converged = sh.converged; r->failed = failed;
    }
}

// ---------------------------------------------------------------------------------------------- exact distance fallback
// For trackable source points whose in-gate neighbour was not found on the grid, brute-force the true nearest target
// (the reference exports the raw kd-tree distance whatever the gate).  packed[s] = float_bits(d2) << 32 | index.
__global__ __launch_bounds__(256) void miss_list_kernel(int n_src, const float* __restrict__ sqd, const int* __restrict__ corr, float gate,
                                                        int* __restrict__ miss, int* __restrict__ n_miss) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_src) return;
    const bool found = sqd[s] < gate;   // exact whenever below the gate (grid completeness radius)
    if (!found) {
        const int k = atomicAdd(n_miss, 1);
        miss[k] = s;
    }
    (void)corr;
}
// One wave per missed query: lanes stride over the (cell-sorted) trackable targets, lexicographic wave-min at the end.
__global__ __launch_bounds__(256) void brute_nn_kernel(const int* __restrict__ miss, const int* __restrict__ n_miss_p, const int* __restrict__ src_track,
                                                       const float4* __restrict__ src_pts, const double* __restrict__ lin_pose,
                                                       const float4* __restrict__ sorted, int n_tgt, float* __restrict__ sqd) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= *n_miss_p) return;   // wave-uniform
    const int s = miss[qi];
    const float4 p = src_pts[src_track[s]];
    float Rf[9], tf[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rf[i] = (float)lin_pose[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tf[i] = (float)lin_pose[9 + i];
    const float qx = ((Rf[0] * p.x + Rf[1] * p.y) + Rf[2] * p.z) + tf[0];
    const float qy = ((Rf[3] * p.x + Rf[4] * p.y) + Rf[5] * p.z) + tf[1];
    const float qz = ((Rf[6] * p.x + Rf[7] * p.y) + Rf[8] * p.z) + tf[2];
    float bd = FLT_MAX;
    int bi = 0x7fffffff;
#pragma unroll 4
    for (int j = lane; j < n_tgt; j += 64) {
        const float4 t = sorted[j];
        const float d = dist2(qx, qy, qz, t.x, t.y, t.z);
        const int id = __float_as_int(t.w);
        if (lex_less(d, id, bd, bi)) { bd = d; bi = id; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (lex_less(od, oi, bd, bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) sqd[s] = bd;
}

// ---------------------------------------------------------------------------------------------- host object
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        size_t want = n + n / 4 + 64;
        if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) { cap = 0; return -1; }
        cap = want;
        return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct Cloud {
    int n = 0, n_track = 0;
    DevBuf<float4> pts;
    DevBuf<int> track;
    DevBuf<double> cov;
    DevBuf<float> rotq, scales;
    bool cov_valid = false, qs_valid = false;
};

}  // namespace

struct gsicp_gicp {
    hipStream_t stream = nullptr;
    int k = 20, max_iter = 64, lm_max_iter = 10, reg = 3;
    double max_corr = (double)FLT_MAX, max_knn = (double)FLT_MAX, rot_eps = 2e-3, trans_eps = 5e-4, lm_init = 1e-9;
    Cloud src, tgt;
    // target search structure
    bool grid_valid = false;
    GridView grid{};
    DevBuf<unsigned long long> gkeys, gskeys, tkeys, tvals, packed;
    DevBuf<unsigned> gvals, gsvals;
    DevBuf<float4> sorted;
    DevBuf<char> sort_temp;
    // per-source-point outputs
    DevBuf<int> corr, miss, counters, nbr_idx;
    DevBuf<float> nbr_d2;
    DevBuf<float> sqd;
    DevBuf<double> maha;
    DevBuf<AlignResult> result;
    DevBuf<AlignSync> sync;
    AlignResult host_result{};
    bool aligned = false, dist_exact = false;
    std::vector<float> h_stage;
    double stats[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

int upload_points(gsicp_gicp* g, Cloud& c, const void* pts, int n, int is_f64) {
    if (n < 0 || (n > 0 && !pts)) { g_last_error = "bad point array"; return -2; }
    c.n = n; c.n_track = n; c.cov_valid = false; c.qs_valid = false;
    if (c.pts.ensure((size_t)n) || c.track.ensure((size_t)n)) { g_last_error = "hipMalloc failed"; return -1; }
    std::vector<float4> h((size_t)n);
    std::vector<int> tr((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (is_f64) { const double* p = (const double*)pts + 3 * (size_t)i; h[i] = make_float4((float)p[0], (float)p[1], (float)p[2], 0.f); }
        else { const float* p = (const float*)pts + 3 * (size_t)i; h[i] = make_float4(p[0], p[1], p[2], 0.f); }
        tr[i] = i;
    }
    if (n > 0) {
        GC(hipMemcpyAsync(c.pts.p, h.data(), sizeof(float4) * n, hipMemcpyHostToDevice, g->stream));
        GC(hipMemcpyAsync(c.track.p, tr.data(), sizeof(int) * n, hipMemcpyHostToDevice, g->stream));
        GC(hipStreamSynchronize(g->stream));
    }
    return 0;
}

int upload_filter(gsicp_gicp* g, Cloud& c, int n_track, const int32_t* f, int n) {
    if (n != c.n) { g_last_error = "filter length does not match the point cloud"; return -2; }
    std::vector<int> tr((size_t)(n_track > 0 ? n_track : 0), -1);
    for (int i = 0; i < n; ++i)
        if (f[i] > 0 && f[i] <= n_track) tr[f[i] - 1] = i;
    size_t w = 0;
    for (size_t i = 0; i < tr.size(); ++i)
        if (tr[i] >= 0) tr[w++] = tr[i];
    c.n_track = (int)w;
    if (c.track.ensure(w ? w : 1)) { g_last_error = "hipMalloc failed"; return -1; }
    if (w) {
        GC(hipMemcpyAsync(c.track.p, tr.data(), sizeof(int) * w, hipMemcpyHostToDevice, g->stream));
        GC(hipStreamSynchronize(g->stream));
    }
    return 0;
}

int calc_cov(gsicp_gicp* g, Cloud& c) {
    const int n = c.n;
    if (c.cov.ensure((size_t)6 * (n ? n : 1)) || c.rotq.ensure((size_t)4 * (n ? n : 1)) || c.scales.ensure((size_t)3 * (n ? n : 1))) {
        g_last_error = "hipMalloc failed"; return -1;
    }
    if (n > 0) {
        if (g->k > 64) { g_last_error = "correspondence randomness (k) > 64 is not supported"; return -2; }
        const float maxd2 = g->max_knn >= (double)FLT_MAX ? FLT_MAX : (float)(g->max_knn * g->max_knn);
        gsicp::ProfileScope ps(gsicp::ST_GICP_COV, g->stream);
        const int nb = (n + 63) / 64;
        int stride = 1;
        for (int p : {37, 41, 43, 47, 53, 59, 61, 67, 71, 73})
            if (p < nb && nb % p != 0) { stride = p; break; }
        if (g->nbr_idx.ensure((size_t)n * 64) || g->nbr_d2.ensure((size_t)n * 64)) { g_last_error = "hipMalloc failed"; return -1; }
        hipLaunchKernelGGL(knn_cov_kernel, dim3((n + 3) / 4), dim3(256), 0, g->stream, n, g->k, stride, c.pts.p, g->nbr_idx.p, g->nbr_d2.p);
        hipLaunchKernelGGL(cov_eig_kernel, dim3((n + 63) / 64), dim3(64), 0, g->stream, n, g->k, c.pts.p, g->nbr_idx.p, g->nbr_d2.p, maxd2,
                           g->reg, c.cov.p, c.rotq.p, c.scales.p);
        GC(hipGetLastError());
    }
    c.cov_valid = true; c.qs_valid = true;
    return 0;
}

int build_grid(gsicp_gicp* g) {
    Cloud& t = g->tgt;
    const int n = t.n_track;
    GridView& G = g->grid;
    std::memset(&G, 0, sizeof(G));
    if (g->sorted.ensure((size_t)(n ? n : 1))) { g_last_error = "hipMalloc failed"; return -1; }
    G.sorted = g->sorted.p; G.n_sorted = n;
    gsicp::ProfileScope ps(gsicp::ST_GICP_GRID, g->stream);
    const bool gated = g->max_corr < 1e6 && g->max_corr > 0;
    if (!gated || n == 0) {
        G.use_grid = 0;
        if (n > 0) hipLaunchKernelGGL(gather_track_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, t.track.p, t.pts.p, g->sorted.p);
        GC(hipGetLastError());
        g->grid_valid = true;
        return 0;
    }
    const float h = (float)(2.0 * g->max_corr * 1.001);
    G.use_grid = 1; G.inv_h = 1.0f / h;
    size_t cap = 64;
    while (cap < (size_t)2 * n) cap <<= 1;
    G.mask = (unsigned)(cap - 1);
    size_t temp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, (size_t)n, 0, 63, g->stream);
    if (g->gkeys.ensure(n) || g->gskeys.ensure(n) || g->gvals.ensure(n) || g->gsvals.ensure(n) || g->tkeys.ensure(cap) ||
        g->tvals.ensure(cap) || g->sort_temp.ensure(temp_bytes ? temp_bytes : 1)) { g_last_error = "hipMalloc failed"; return -1; }
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(grid_keys_kernel, grid, block, 0, g->stream, n, t.track.p, t.pts.p, G.inv_h, g->gkeys.p, g->gvals.p);
    GC(rocprim::radix_sort_pairs(g->sort_temp.p, temp_bytes, g->gkeys.p, g->gskeys.p, g->gvals.p, g->gsvals.p, (size_t)n, 0, 63, g->stream));
    GC(hipMemsetAsync(g->tkeys.p, 0xFF, cap * 8, g->stream));
    GC(hipMemsetAsync(g->tvals.p, 0, cap * 8, g->stream));
    hipLaunchKernelGGL(grid_fill_kernel, grid, block, 0, g->stream, n, g->gskeys.p, g->gsvals.p, t.pts.p, g->sorted.p, G.mask, g->tkeys.p, g->tvals.p);
    GC(hipGetLastError());
    G.keys = g->tkeys.p; G.vals = g->tvals.p;
    g->grid_valid = true;
    return 0;
}

int fetch_floats(gsicp_gicp* g, const float* dev, int n_pts, int width, float* out, int cap_pts) {
    const int n = n_pts < cap_pts ? n_pts : cap_pts;
    if (n > 0) {
        GC(hipMemcpyAsync(out, dev, sizeof(float) * (size_t)n * width, hipMemcpyDeviceToHost, g->stream));
        GC(hipStreamSynchronize(g->stream));
    }
    return n;
}

}  // namespace

extern "C" {

gsicp_gicp* gsicp_gicp_create(void) {
    gsicp_gicp* g = new gsicp_gicp();
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
        g_last_error = "hipStreamCreate failed (no HIP device?)";
        delete g;
        return nullptr;
    }
    if (g->result.ensure(1) || g->counters.ensure(4) || g->sync.ensure(1)) { g_last_error = "hipMalloc failed"; delete g; return nullptr; }
    return g;
}
void gsicp_gicp_destroy(gsicp_gicp* g) {
    if (!g) return;
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    delete g;
}
int gsicp_gicp_set_max_correspondence_distance(gsicp_gicp* g, double d) { g->max_corr = d; g->grid_valid = false; return 0; }
int gsicp_gicp_set_max_knn_distance(gsicp_gicp* g, double d) { g->max_knn = d; g->src.cov_valid = false; return 0; }
int gsicp_gicp_set_correspondence_randomness(gsicp_gicp* g, int k) {
    if (k < 1 || k > 64) { g_last_error = "k must be in [1, 64]"; return -2; }
    g->k = k; return 0;
}
int gsicp_gicp_set_max_iterations(gsicp_gicp* g, int n) { g->max_iter = n; return 0; }
int gsicp_gicp_set_num_threads(gsicp_gicp*, int) { return 0; }
int gsicp_gicp_set_regularization_method(gsicp_gicp* g, int m) {
    if (m < 0 || m > 4) { g_last_error = "unknown regularization method"; return -2; }
    g->reg = m; g->src.cov_valid = false; return 0;
}
int gsicp_gicp_set_rotation_epsilon(gsicp_gicp* g, double e) { g->rot_eps = e; return 0; }
int gsicp_gicp_set_transformation_epsilon(gsicp_gicp* g, double e) { g->trans_eps = e; return 0; }

int gsicp_gicp_set_input_target(gsicp_gicp* g, const void* pts, int n, int is_f64) {
    g->grid_valid = false; g->aligned = false;
    return upload_points(g, g->tgt, pts, n, is_f64);
}
int gsicp_gicp_set_input_source(gsicp_gicp* g, const void* pts, int n, int is_f64) {
    g->aligned = false;
    return upload_points(g, g->src, pts, n, is_f64);
}
int gsicp_gicp_set_target_filter(gsicp_gicp* g, int n_track, const int32_t* f, int n) {
    g->grid_valid = false;
    return upload_filter(g, g->tgt, n_track, f, n);
}
int gsicp_gicp_set_source_filter(gsicp_gicp* g, int n_track, const int32_t* f, int n) { return upload_filter(g, g->src, n_track, f, n); }
int gsicp_gicp_calculate_target_covariance_with_filter(gsicp_gicp* g) {
    const int rc = calc_cov(g, g->tgt);
    if (rc == 0) GC(hipStreamSynchronize(g->stream));
    return rc;
}
int gsicp_gicp_calculate_source_covariance(gsicp_gicp* g) {
    const int rc = calc_cov(g, g->src);
    if (rc == 0) GC(hipStreamSynchronize(g->stream));
    return rc;
}
int gsicp_gicp_get_target_rotationsq(gsicp_gicp* g, float* out, int cap) {
    if (!g->tgt.qs_valid) { g_last_error = "target covariances have not been computed"; return -2; }
    return fetch_floats(g, g->tgt.rotq.p, g->tgt.n, 4, out, cap);
}
int gsicp_gicp_get_target_scales(gsicp_gicp* g, float* out, int cap) {
    if (!g->tgt.qs_valid) { g_last_error = "target covariances have not been computed"; return -2; }
    return fetch_floats(g, g->tgt.scales.p, g->tgt.n, 3, out, cap);
}
int gsicp_gicp_get_source_rotationsq(gsicp_gicp* g, float* out, int cap) {
    if (!g->src.qs_valid) { if (int rc = calc_cov(g, g->src)) return rc; }
    return fetch_floats(g, g->src.rotq.p, g->src.n, 4, out, cap);
}
int gsicp_gicp_get_source_scales(gsicp_gicp* g, float* out, int cap) {
    if (!g->src.qs_valid) { if (int rc = calc_cov(g, g->src)) return rc; }
    return fetch_floats(g, g->src.scales.p, g->src.n, 3, out, cap);
}
int gsicp_gicp_set_target_covariances_fromqs(gsicp_gicp* g, const float* rots, int n_rots, const float* scales, int n_scales) {
    Cloud& t = g->tgt;
    if (n_rots != 4 * t.n || n_scales != 3 * t.n) { g_last_error = "rotations / scales do not match the target cloud size"; return -2; }
    const size_t n = (size_t)(t.n ? t.n : 1);
    if (t.cov.ensure(6 * n) || t.rotq.ensure(4 * n) || t.scales.ensure(3 * n)) { g_last_error = "hipMalloc failed"; return -1; }
    if (t.n > 0) {
        GC(hipMemcpyAsync(t.rotq.p, rots, sizeof(float) * 4 * t.n, hipMemcpyHostToDevice, g->stream));
        GC(hipMemcpyAsync(t.scales.p, scales, sizeof(float) * 3 * t.n, hipMemcpyHostToDevice, g->stream));
        hipLaunchKernelGGL(cov_fromqs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, g->stream, t.n, g->reg, t.rotq.p, t.scales.p, t.cov.p);
        GC(hipGetLastError());
        GC(hipStreamSynchronize(g->stream));
    }
    t.cov_valid = true; t.qs_valid = true;
    return 0;
}

int gsicp_gicp_align(gsicp_gicp* g, const double* init, double* out) {
    Cloud &s = g->src, &t = g->tgt;
    if (s.n == 0 || t.n == 0) { g_last_error = "align: source and target must be set"; return -2; }
    hipEvent_t e0, e1;
    GC(hipEventCreate(&e0)); GC(hipEventCreate(&e1));
    GC(hipEventRecord(e0, g->stream));
    int launches = 0;
    if (!s.cov_valid) { if (int rc = calc_cov(g, s)) return rc; ++launches; }
    if (!t.cov_valid) { if (int rc = calc_cov(g, t)) return rc; ++launches; }
    if (!g->grid_valid) { if (int rc = build_grid(g)) return rc; launches += 4; }
    const size_t ns = (size_t)(s.n_track ? s.n_track : 1);
    if (g->corr.ensure(ns) || g->sqd.ensure(ns) || g->maha.ensure(6 * ns) || g->miss.ensure(ns) || g->packed.ensure(ns)) {
        g_last_error = "hipMalloc failed"; return -1;
    }
    AlignArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_src = s.n_track; a.src_track = s.track.p; a.src_pts = s.pts.p; a.src_cov = s.cov.p;
    a.tgt_pts = t.pts.p; a.tgt_cov = t.cov.p; a.grid = g->grid;
    a.gate = g->max_corr >= 1e18 ? FLT_MAX : (float)g->max_corr * (float)g->max_corr;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) a.init[3 * r + c] = (double)(float)init[4 * r + c];
        a.init[9 + r] = (double)(float)init[4 * r + 3];
    }
    a.max_iter = g->max_iter; a.lm_max_iter = g->lm_max_iter; a.rot_eps = g->rot_eps; a.trans_eps = g->trans_eps; a.lm_init = g->lm_init;
    a.corr = g->corr.p; a.sqd = g->sqd.p; a.maha = g->maha.p; a.result = g->result.p; a.sync = g->sync.p;
    int nwg = (s.n_track + AL_T - 1) / AL_T;
    if (nwg < 1) nwg = 1;
    if (nwg > AL_MAX_WG) nwg = AL_MAX_WG;
    GC(hipMemsetAsync(g->sync.p, 0, 128, g->stream));   // barrier counter + abort flag
    { gsicp::ProfileScope ps(gsicp::ST_GICP_ALIGN, g->stream);
      hipLaunchKernelGGL(gicp_align_kernel, dim3(nwg), dim3(AL_T), 0, g->stream, a); }
    ++launches;
    GC(hipGetLastError());
    GC(hipEventRecord(e1, g->stream));
    GC(hipMemcpyAsync(&g->host_result, g->result.p, sizeof(AlignResult), hipMemcpyDeviceToHost, g->stream));
    GC(hipStreamSynchronize(g->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::memcpy(out, g->host_result.final_pose, sizeof(double) * 16);
    g->aligned = true; g->dist_exact = false;
    g->stats[0] = launches; g->stats[1] = g->host_result.lm_trials; g->stats[2] = g->host_result.cost;
    g->stats[3] = g->host_result.converged; g->stats[4] = ms * 1000.0; g->stats[5] = g->host_result.failed;
    return g->host_result.iterations;
}

int gsicp_gicp_get_source_correspondence(gsicp_gicp* g, int32_t* idx, float* d2, int cap) {
    if (!g->aligned) { g_last_error = "get_source_correspondence before align"; return -2; }
    Cloud &s = g->src, &t = g->tgt;
    const int n = s.n_track;
    if (!g->dist_exact && g->grid.use_grid && n > 0 && t.n_track > 0) {
        const float gate = (float)g->max_corr * (float)g->max_corr;
        gsicp::ProfileScope ps(gsicp::ST_GICP_MISS, g->stream);
        GC(hipMemsetAsync(g->counters.p, 0, sizeof(int) * 4, g->stream));
        hipLaunchKernelGGL(miss_list_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, g->sqd.p, g->corr.p, gate, g->miss.p,
                           g->counters.p);
        hipLaunchKernelGGL(brute_nn_kernel, dim3((n + 3) / 4), dim3(256), 0, g->stream, g->miss.p, g->counters.p, s.track.p, s.pts.p,
                           g->result.p->lin_pose, g->sorted.p, t.n_track, g->sqd.p);
        GC(hipGetLastError());
        g->dist_exact = true;
    }
    const int m = n < cap ? n : cap;
    if (m > 0) {
        GC(hipMemcpyAsync(idx, g->corr.p, sizeof(int) * m, hipMemcpyDeviceToHost, g->stream));
        GC(hipMemcpyAsync(d2, g->sqd.p, sizeof(float) * m, hipMemcpyDeviceToHost, g->stream));
    }
    GC(hipStreamSynchronize(g->stream));
    return m;
}
int gsicp_gicp_num_source(gsicp_gicp* g) { return g->src.n; }
int gsicp_gicp_num_target(gsicp_gicp* g) { return g->tgt.n; }
int gsicp_gicp_last_align_stats(gsicp_gicp* g, double out[6]) { std::memcpy(out, g->stats, sizeof(double) * 6); return 0; }
int gsicp_gicp_get_final_hessian(gsicp_gicp* g, double out[36]) { std::memcpy(out, g->host_result.H_final, sizeof(double) * 36); return 0; }

}  // extern "C"
