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
//  * One align() = ONE kernel launch: a persistent grid (<= 240 workgroups, one source point per thread) runs every
//    Levenberg-Marquardt iteration on the device (correspondences, Mahalanobis matrices, 6x6 normal equations reduced in
//    fp64 with permlane-swap folding + LDS + a grid barrier, LDL^T solve, SE(3) exponential, trial cost, accept/reject).
//    The trial phase also linearises at the trial pose, so an accepted step costs one grid-wide phase, not two.  The result
//    reaches the host through a pinned mailbox the kernel writes: no read-back copy, no stream synchronise.
//  * k-NN covariances (k = 20 over <= 12 k points): counting sort of the cloud into a uniform grid on the device, then one
//    wave per query with the running top-k held across the lanes, cells visited nearest-first and pruned against the k-th
//    distance, exactness guaranteed by the distance to the faces of the scanned cube (ring growth, full scan as last resort);
//    the fp64 covariance / Jacobi eigen-decomposition runs one thread per point in a second kernel.
//  * Per-point arithmetic follows the reference algorithm's precision: correspondence search in fp32 with a fixed
//    evaluation order (this file is compiled with -ffp-contract=off, so nearest-neighbour indices are bit-exact
//    against the CPU oracle), cost / Jacobians / reductions in fp64.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
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
    float inv_hf, hf;         // FINE cell edge hf = gate radius x 1.001 and its inverse; a hashed COARSE cell is a 2x2x2 block of fine cells
    unsigned mask;            // table capacity - 1
    const unsigned long long* keys;   // coarse-cell key per slot (EMPTY_KEY = free); probe sequences start at 4-slot buckets
    const unsigned* ords;             // per slot: ordinal of the coarse cell
    const uint2* cells;               // [ordinal * 8 + octant] = {first index in `sorted`, count} of that fine cell
    const float4* sorted;     // xyz + original index bits, sorted by (coarse cell, octant)
    int n_sorted;
};

// ---------------------------------------------------------------------------------------------- device helpers
// 20 bits per axis, so that key << 3 | octant still fits the 63 bits the radix sort looks at: +-20 km with the gate-sized level's coarse cells
// (2 x gate >= 4 cm), +-1.5 km with the second level's smallest cells (2 x 0.15 x gate = 6 mm at Replica's gate).  Beyond that the key wraps:
// two far-apart cells then share a key and a query scans both — slower, never wrong (every candidate's distance is computed exactly).
__device__ inline unsigned long long cell_key(int cx, int cy, int cz) {
    const unsigned long long o = 1ull << 19;
    return ((unsigned long long)(cx + (long long)o) & 0xFFFFFull) << 40 | ((unsigned long long)(cy + (long long)o) & 0xFFFFFull) << 20 |
           ((unsigned long long)(cz + (long long)o) & 0xFFFFFull);
}
__device__ inline unsigned hash_key(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}
__device__ inline float dist2(float qx, float qy, float qz, float px, float py, float pz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    return dx * dx + dy * dy + dz * dz;
}
constexpr unsigned NO_CELL = 0xFFFFFFFFu;

// Exact nearest neighbour among the 27 FINE cells around q (complete for every target point within the gate radius of q: a fine cell's
// edge is the radius x 1.001).  Ties -> lowest original index.  Returns best squared distance / index (FLT_MAX / -1 if the cells are empty).
//
// Round 4: map-sized targets.  Until round 3 the index was the coarse grid alone (cell edge = 2 x gate, the 8 cells around q scanned in
// full): right for a depth frame's cloud (1.5 points per cell), but the tracker's steady-state target is the MAP — 1e5..1e6 Gaussians on the
// same surfaces, 6..60 points per cell (max 257 measured) — and one lane walking 8 x 57 candidates four loads at a time made a linearisation
// 200 us (align 655 us at K = 1e6 against 64 us at K = 8 k, profiles/r04_tracker_vs_map_before_octants.json).  Now the points of a coarse cell are
// sorted by OCTANT (2x2x2 fine cells) and every coarse cell carries the eight {begin, count} pairs: a query visits its own fine cell first,
// then the 26 around it nearest-first, and skips every fine cell whose box is farther than the best distance so far (strictly: ties still
// scan, the tie-break is by index).  In a dense map the own cell settles it; the result is the same bits: (d, id) is a total order and
// a skipped cell cannot hold a smaller pair.  The 27 fine cells live in exactly the 8 coarse cells the old search hashed, so the table
// probes are unchanged: ONE round of independent bucket loads, one round of 27 {begin, count} loads, then the candidates.
constexpr int AL_T_CONST = 256;   // = AL_T (threads per workgroup of the align kernel); stride of the per-lane LDS work list
__device__ __forceinline__ void grid_nn(const GridView& g, float qx, float qy, float qz, float& best_d, int& best_i, uint2* __restrict__ queue) {
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
    const float flx = floorf(qx * g.inv_hf), fly = floorf(qy * g.inv_hf), flz = floorf(qz * g.inv_hf);
    const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
    const int px = ix & 1, py = iy & 1, pz = iz & 1;              // which half of its (home) coarse cell the query's fine cell is
    const int hx = ix >> 1, hy = iy >> 1, hz = iz >> 1;           // home coarse cell (arithmetic shift = floor division)
    const int sx = px ? 1 : -1, sy = py ? 1 : -1, sz = pz ? 1 : -1;   // the other coarse cell on each axis lies on the side of the query's half
    // ---- round 1: the 8 coarse cells, one 4-slot bucket each (4 keys + 4 ordinals = three 16-byte loads), all in flight together.  A key's
    // probe sequence starts at its bucket; the sequential continuation only runs when a bucket is full of other keys (load factor <= 1/8 in
    // POINTS, far less in cells).  Latency is everything here: one point per thread, one wave per SIMD.
    unsigned long long key[8];
    unsigned slot[8], ord[8];
    ulonglong2 ka[8], kb[8];
    uint4 oa[8];
    const ulonglong2* K2 = (const ulonglong2*)g.keys;
    const uint4* O4 = (const uint4*)g.ords;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        key[c] = cell_key(hx + ((c & 1) ? sx : 0), hy + ((c & 2) ? sy : 0), hz + ((c & 4) ? sz : 0));
        slot[c] = (hash_key(key[c]) << 2) & g.mask;     // first slot of the key's bucket
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned h2 = slot[c] >> 1;
        ka[c] = K2[h2]; kb[c] = K2[h2 + 1]; oa[c] = O4[slot[c] >> 2];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned long long k = key[c];
        unsigned v = NO_CELL;
        if (ka[c].x == k) v = oa[c].x;
        else if (ka[c].x == EMPTY_KEY) v = NO_CELL;
        else if (ka[c].y == k) v = oa[c].y;
        else if (ka[c].y == EMPTY_KEY) v = NO_CELL;
        else if (kb[c].x == k) v = oa[c].z;
        else if (kb[c].x == EMPTY_KEY) v = NO_CELL;
        else if (kb[c].y == k) v = oa[c].w;
        else if (kb[c].y != EMPTY_KEY) {          // bucket full of other keys: continue the linear probe
            unsigned sl = (slot[c] + 4) & g.mask;
            for (;;) {
                const unsigned long long kk = g.keys[sl];
                if (kk == k) { v = g.ords[sl]; break; }
                if (kk == EMPTY_KEY) break;
                sl = (sl + 1) & g.mask;
            }
        }
        ord[c] = v;
    }
    // ---- round 2: {begin, count} of the 27 fine cells.  Fine cell (dx, dy, dz) relative to the query's: on each axis d = 0 and d = -s stay in
    // the home coarse cell (octant bit p and p ^ 1), d = +s is the other coarse cell's adjacent half (octant bit p ^ 1).  The coarse ordinal
    // is picked by three levels of selects shared between the cells (38 selects, not 27 x 7); everything is indexed statically.
    unsigned A[3][2][2], B[3][3][2], C[3][3][3];
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
            const unsigned o0 = ord[0 + 2 * cy + 4 * cz], o1 = ord[1 + 2 * cy + 4 * cz];
            A[1][cy][cz] = o0;
            A[2][cy][cz] = px ? o1 : o0;     // dx = +1 is the other coarse cell iff the query sits in the upper half
            A[0][cy][cz] = px ? o0 : o1;
        }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
            B[dx][1][cz] = A[dx][0][cz];
            B[dx][2][cz] = py ? A[dx][1][cz] : A[dx][0][cz];
            B[dx][0][cz] = py ? A[dx][0][cz] : A[dx][1][cz];
        }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            C[dx][dy][1] = B[dx][dy][0];
            C[dx][dy][2] = pz ? B[dx][dy][1] : B[dx][dy][0];
            C[dx][dy][0] = pz ? B[dx][dy][0] : B[dx][dy][1];
        }
    const unsigned obx[3] = {(unsigned)(px ^ 1), (unsigned)px, (unsigned)(px ^ 1)};
    const unsigned oby[3] = {(unsigned)(py ^ 1) << 1, (unsigned)py << 1, (unsigned)(py ^ 1) << 1};
    const unsigned obz[3] = {(unsigned)(pz ^ 1) << 2, (unsigned)pz << 2, (unsigned)(pz ^ 1) << 2};
    uint2 rec[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const int dx = t % 3, dy = (t / 3) % 3, dz = t / 9;
        const unsigned o = C[dx][dy][dz];
        const unsigned idx = (o == NO_CELL ? 0u : o * 8u) + (obx[dx] + oby[dy] + obz[dz]);   // unconditional load (cell 0 exists when the grid is in use)
        rec[t] = g.cells[idx];
        if (o == NO_CELL) rec[t].y = 0u;
    }
    // squared distance from q to the slab of fine cells at offset -1 / +1 on each axis, made CONSERVATIVE: the cell of a point is
    // floorf(p * inv_hf) in float arithmetic, so a cell's real extent is blurred by a few ulps of the coordinate, and the candidates'
    // own distances are rounded — the slack (a thousandth of a cell, or 8 ulps of the coordinate if larger) only ever makes a cell look
    // nearer, i.e. scans a little more.
    float gm2[3], gp2[3];
    {
        const float q[3] = {qx, qy, qz}, fl[3] = {flx, fly, flz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float slack = fmaxf(g.hf * 1e-3f, fabsf(q[a]) * 9.5367431640625e-07f);
            const float lo = fl[a] * g.hf, hi = (fl[a] + 1.0f) * g.hf;
            const float gm = fmaxf((q[a] - lo) - slack, 0.f), gp = fmaxf((hi - q[a]) - slack, 0.f);
            gm2[a] = gm * gm * 0.9999f; gp2[a] = gp * gp * 0.9999f;
        }
    }
    // ---- round 3+: own fine cell first, then the non-empty neighbours nearest-first (faces, edges, corners), each skipped when its box is
    // farther than the best pair so far.  The neighbours wait in a per-lane work list in LDS (`queue[k * AL_T]`, filled by static code; an
    // entry packs the cell's position code, 2 bits per axis, above the count < 2^26), and ONE loop scans: every lane walks ITS OWN list —
    // skipping pruned entries costs LDS reads only — and takes 8 candidates per trip (8 independent loads: 128 B, one cache line when
    // aligned), so a wave makes as many trips as its busiest lane has 8-candidate chunks, not one per list slot anybody uses.  (26 statically
    // unrolled scan loops made the persistent kernel 17 k instructions: it stopped being inlined as a whole and its fp64 accumulators went
    // through scratch memory.)
    int nq = 0;
#pragma unroll
    for (int pass = 1; pass < 4; ++pass) {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int dx = t % 3, dy = (t / 3) % 3, dz = t / 9;
            if ((dx != 1) + (dy != 1) + (dz != 1) != pass) continue;
            if (rec[t].y != 0u) {
                queue[nq * AL_T_CONST] = make_uint2(rec[t].x, rec[t].y | (unsigned)(dx | dy << 2 | dz << 4) << 26);
                ++nq;
            }
        }
    }
    unsigned cs = rec[13].x, cc = rec[13].y, cj = 0u;     // the range being scanned (starts with the own cell), offset of its next chunk
    int qi = 0;
    for (;;) {
        while (cj >= cc && qi < nq) {
            const uint2 e = queue[qi * AL_T_CONST];
            ++qi;
            const unsigned code = e.y >> 26;
            const unsigned cx = code & 3u, cy = (code >> 2) & 3u, cz = code >> 4;
            const float m2 = (cx == 0u ? gm2[0] : (cx == 2u ? gp2[0] : 0.f)) + (cy == 0u ? gm2[1] : (cy == 2u ? gp2[1] : 0.f)) +
                             (cz == 0u ? gm2[2] : (cz == 2u ? gp2[2] : 0.f));
            if (m2 > best_d) continue;
            cs = e.x; cc = e.y & 0x3FFFFFFu; cj = 0u;
        }
        if (cj >= cc) break;
        // 8 candidates per trip, 16 where more than 8 are left in the range (the second eight sit behind a wave-uniform skip: a sparse
        // target never issues them, a dense cell halves its trips — every trip is a dependent memory latency)
        float4 pj[16];
        const bool wide = cc - cj > 8u;
#pragma unroll
        for (int u = 0; u < 8; ++u) pj[u] = g.sorted[cs + (cj + u < cc ? cj + u : cj)];
        if (wide) {
#pragma unroll
            for (int u = 8; u < 16; ++u) pj[u] = g.sorted[cs + (cj + u < cc ? cj + u : cj)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cj + u < cc) {
                const float d = dist2(qx, qy, qz, pj[u].x, pj[u].y, pj[u].z);
                const int id = __float_as_int(pj[u].w);
                if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; }
            }
        }
        if (wide) {
#pragma unroll
            for (int u = 8; u < 16; ++u) {
                if (cj + u < cc) {
                    const float d = dist2(qx, qy, qz, pj[u].x, pj[u].y, pj[u].z);
                    const int id = __float_as_int(pj[u].w);
                    if (d < best_d || (d == best_d && id < best_i)) { best_d = d; best_i = id; }
                }
            }
        }
        cj += wide ? 16u : 8u;
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
// The 64 lanes take candidates 64 at a time (one coalesced 1 KiB load per cell range); the running top-k is a sorted
// list held ACROSS the lanes (lane r = rank r), so an insertion is a ballot + popcount + one wave_shr DPP move
// instead of a 20-deep per-thread compare chain.  Ordering is lexicographic (d2, index) like the oracle's.
__device__ inline bool lex_less(float d, int i, float d2, int i2) { return d < d2 || (d == d2 && i < i2); }
__device__ inline float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }  // lane l <- lane l-1

// ---- exact k-NN over a uniform grid -----------------------------------------------------------------------------
// The brute-force scan above costs n/64 batches per query (130 at n = 8 280).  A counting sort of the cloud into grid
// cells (dense cell_start table, all on the device) lets a query look only at the (2r+1)^3 cells around it: each (y,z) row
// of that cube is ONE contiguous range of the sorted array.  The answer stays exact: after ring r the k-th distance must be
// below the distance to the nearest face of the scanned cube, otherwise the ring grows (r <= KNN_RMAX), and past that the
// query falls back to scanning every point.  The result does not depend on the visiting order (total order on (d2, index)).
constexpr int KNN_MAX_CELLS = 1 << 18;
#ifndef KNN_H_AREA
#define KNN_H_AREA 2.25f                    // cell edge in units of the estimated point spacing (surface / volume model).  Round 6 sweep on the steady-state frame
                                            // (profiles/r06_knn_cell_size_sweep.json; the search is exact for any value): 1.0 .. 1.75 = 174 / 112 / 89 / 84 us of
                                            // k-NN + covariances (second rings), 2.0 = 45.3, 2.25 = 45.8, 2.5 = 47.3 (rounds 1-5), 3.0 = 50.3: 2.25 keeps a margin to the cliff
#define KNN_H_VOL 1.26f                     // (the same ratio to KNN_H_AREA as rounds 1-5: 1.4 / 2.5)
#endif
struct KnnGrid {
    float ox, oy, oz, h, inv_h;
    int nx, ny, nz, ncells;
    unsigned ring_hist[5];   // diagnostics: queries settled by whole-grid coverage [0], after ring 1 / 2 [1], [2], by the full scan [4]
};

// grid parameters from the bounding box (one definition for every builder below: same arithmetic, same grid)
__device__ inline KnnGrid knn_grid_from_box(const float L[3], const float E[3], int n, float h_area, float h_vol) {
    const float emax = fmaxf(fmaxf(E[0], E[1]), fmaxf(E[2], 1e-6f));
    // depth-camera clouds are surfaces: spacing ~ sqrt(area / n); 20 neighbours sit within ~2.5 spacings, and a
    // cell edge of 4 leaves room for the density to vary across the cloud before a second ring is needed.  The volume term covers
    // genuinely volumetric clouds.  Either way the search below is exact; h only decides how much it scans.
    const float area = E[0] * E[1] + E[1] * E[2] + E[0] * E[2];
    const float vol = fmaxf(E[0], 1e-3f * emax) * fmaxf(E[1], 1e-3f * emax) * fmaxf(E[2], 1e-3f * emax);
    float h = fmaxf(h_area * sqrtf(area / (float)n), h_vol * cbrtf(vol / (float)n));
    h = fmaxf(h, emax * (1.f / 1024.f));
    int nx, ny, nz;
    for (;;) {
        nx = (int)(E[0] / h) + 1; ny = (int)(E[1] / h) + 1; nz = (int)(E[2] / h) + 1;
        if ((long long)nx * ny * nz <= KNN_MAX_CELLS) break;
        h *= 1.25f;
    }
    KnnGrid g;
    g.ox = L[0]; g.oy = L[1]; g.oz = L[2]; g.h = h; g.inv_h = 1.f / h;
    g.nx = nx; g.ny = ny; g.nz = nz; g.ncells = nx * ny * nz;
    for (int i = 0; i < 5; ++i) g.ring_hist[i] = 0u;
    return g;
}

// Clouds beyond KNN_FUSED_MAX_N points (a map's trackable Gaussians at every tracking keyframe, distCUDA2's new keyframe points, large
// frames): the build as SIX chip-wide launches — partial boxes, parameters + counter reset, count, tile sums, scan, fill.  The scratch of
// the build (partial boxes, tile sums) follows the KnnGrid record in the same allocation (KNN_PARAM_SLOTS records); the search kernels
// only ever see its first record.  Rounds 1-3 ran the box and the scan as ONE workgroup each (52 + 125 us at 1.5e5 points, 350 us of box
// alone at 1e6): half of a tracking keyframe's index build.
constexpr int KNN_BOX_WGS = 256;             // partial boxes
constexpr int KNN_SCAN_TILE = 4096;          // counters per scan workgroup (1024 threads x 4)
constexpr int KNN_SCAN_TILES = KNN_MAX_CELLS / KNN_SCAN_TILE + 1;   // covers counter index ncells (= 0: its exclusive prefix is the total)
constexpr int KNN_CELL_SLOTS = KNN_SCAN_TILES * KNN_SCAN_TILE;      // counter arrays are whole tiles (aligned 16-byte loads of the last one)
constexpr int KNN_PARAM_WGS = 32;
struct KnnBuild {
    KnnGrid g;
    unsigned tile_sum[KNN_SCAN_TILES + 3];
    float part[6][KNN_BOX_WGS];              // lo x/y/z, hi x/y/z per workgroup of knn_box_kernel
};
constexpr int KNN_PARAM_SLOTS = (int)((sizeof(KnnBuild) + sizeof(KnnGrid) - 1) / sizeof(KnnGrid));

__global__ __launch_bounds__(256) void knn_box_kernel(int n, const float4* __restrict__ pts, KnnGrid* __restrict__ gp) {
    KnnBuild* const kb = (KnnBuild*)gp;
    __shared__ float s_v[6][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int stride = (int)gridDim.x * 256;
#pragma unroll 4
    for (int i = (int)blockIdx.x * 256 + tid; i < n; i += stride) {
        const float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
        if (lane == 0) { s_v[d][wave] = lo[d]; s_v[3 + d][wave] = hi[d]; }
    }
    __syncthreads();
    if (tid < 6) {
        const float a = s_v[tid][0], b = s_v[tid][1], c = s_v[tid][2], e = s_v[tid][3];
        kb->part[tid][blockIdx.x] = tid < 3 ? fminf(fminf(a, b), fminf(c, e)) : fmaxf(fmaxf(a, b), fmaxf(c, e));
    }
}
// every workgroup folds the partial boxes (minima and maxima: any order gives the same bits), derives the same parameters and clears its
// share of the cell counters [0, ncells]; workgroup 0 publishes the record
__global__ __launch_bounds__(1024) void knn_grid_params_kernel(int n, int n_parts, float h_area, float h_vol, KnnGrid* __restrict__ gp,
                                                               unsigned* __restrict__ cell_count) {
    const KnnBuild* const kb = (const KnnBuild*)gp;
    __shared__ float s_v[6][16];
    __shared__ int s_ncells;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        float v = d < 3 ? FLT_MAX : -FLT_MAX;
        for (int b = tid; b < n_parts; b += 1024) v = d < 3 ? fminf(v, kb->part[d][b]) : fmaxf(v, kb->part[d][b]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = d < 3 ? fminf(v, __shfl_xor(v, off, 64)) : fmaxf(v, __shfl_xor(v, off, 64));
        if (lane == 0) s_v[d][wave] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float L[3], E[3];
        for (int d = 0; d < 3; ++d) {
            float l = s_v[d][0], u = s_v[3 + d][0];
            for (int w = 1; w < 16; ++w) { l = fminf(l, s_v[d][w]); u = fmaxf(u, s_v[3 + d][w]); }
            L[d] = l; E[d] = fmaxf(u - l, 0.f);
        }
        const KnnGrid g = knn_grid_from_box(L, E, n, h_area, h_vol);
        if (blockIdx.x == 0) *gp = g;
        s_ncells = g.ncells;
    }
    __syncthreads();
    const int ncells = s_ncells;
    for (int c = (int)blockIdx.x * 1024 + tid; c <= ncells; c += (int)gridDim.x * 1024) cell_count[c] = 0u;
}
__device__ inline void knn_cell_of(const KnnGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = (int)((x - g.ox) * g.inv_h); cy = (int)((y - g.oy) * g.inv_h); cz = (int)((z - g.oz) * g.inv_h);
    cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.ny ? g.ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
}
__global__ __launch_bounds__(256) void knn_count_kernel(int n, const float4* __restrict__ pts, const KnnGrid* __restrict__ gp,
                                                        int* __restrict__ cell_of, unsigned* __restrict__ cell_count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = *gp;
    const float4 p = pts[i];
    int cx, cy, cz;
    knn_cell_of(g, p.x, p.y, p.z, cx, cy, cz);
    const int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    atomicAdd(&cell_count[c], 1u);
}
// exclusive scan of the cell counts -> cell_start[0..ncells]; cell_fill (cursor) = copy of cell_start.  One workgroup per tile of 4 096
// counters: tile sums, then every tile adds the sums of the tiles before it (<= 64 values) to its own in-tile scan.
__device__ inline uint4 knn_load_counts(const unsigned* __restrict__ cell_count, int base, int ncells) {
    uint4 v = *(const uint4*)(cell_count + base);     // whole tiles are allocated; entries beyond ncells are not initialised: masked here
    if (base + 0 > ncells) v.x = 0u;
    if (base + 1 > ncells) v.y = 0u;
    if (base + 2 > ncells) v.z = 0u;
    if (base + 3 > ncells) v.w = 0u;
    return v;
}
__global__ __launch_bounds__(1024) void knn_scan_sums_kernel(KnnGrid* __restrict__ gp, const unsigned* __restrict__ cell_count) {
    KnnBuild* const kb = (KnnBuild*)gp;
    __shared__ unsigned s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncells = gp->ncells;
    const int base = (int)blockIdx.x * KNN_SCAN_TILE + 4 * tid;
    unsigned sum = 0u;
    if ((int)blockIdx.x * KNN_SCAN_TILE <= ncells) { const uint4 v = knn_load_counts(cell_count, base, ncells); sum = (v.x + v.y) + (v.z + v.w); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (unsigned)__shfl_xor((int)sum, off, 64);
    if (lane == 0) s_w[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned t = 0u;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        kb->tile_sum[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void knn_scan_kernel(const KnnGrid* __restrict__ gp, const unsigned* __restrict__ cell_count,
                                                        unsigned* __restrict__ cell_start, unsigned* __restrict__ cell_fill) {
    const KnnBuild* const kb = (const KnnBuild*)gp;
    __shared__ unsigned s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncells = gp->ncells;
    if ((int)blockIdx.x * KNN_SCAN_TILE > ncells) return;       // workgroup-uniform
    unsigned before = 0u;                                       // counters of the tiles in front of this one
    for (int b = lane; b < (int)blockIdx.x; b += 64) before += kb->tile_sum[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += (unsigned)__shfl_xor((int)before, off, 64);
    const int base = (int)blockIdx.x * KNN_SCAN_TILE + 4 * tid;
    const uint4 v = knn_load_counts(cell_count, base, ncells);
    const unsigned mine = (v.x + v.y) + (v.z + v.w);
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned run = before + incl - mine;
#pragma unroll
    for (int w = 0; w < 16; ++w) if (w < wave) run += s_w[w];
    const unsigned r0 = run, r1 = r0 + v.x, r2 = r1 + v.y, r3 = r2 + v.z;
    if (base + 3 <= ncells) {
        *(uint4*)(cell_start + base) = make_uint4(r0, r1, r2, r3);
        *(uint4*)(cell_fill + base) = make_uint4(r0, r1, r2, r3);
    } else {
        if (base + 0 <= ncells) { cell_start[base + 0] = r0; cell_fill[base + 0] = r0; }
        if (base + 1 <= ncells) { cell_start[base + 1] = r1; cell_fill[base + 1] = r1; }
        if (base + 2 <= ncells) { cell_start[base + 2] = r2; cell_fill[base + 2] = r2; }
    }
}
__global__ __launch_bounds__(256) void knn_fill_kernel(int n, const float4* __restrict__ pts, const int* __restrict__ cell_of,
                                                       unsigned* __restrict__ cell_fill, float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    p.w = __int_as_float(i);
    sorted[atomicAdd(&cell_fill[cell_of[i]], 1u)] = p;
}

// the multi-launch build of the uniform grid over `pts` (see KnnBuild): gp must hold KNN_PARAM_SLOTS records, the counter arrays KNN_CELL_SLOTS
inline void enqueue_knn_grid_build(hipStream_t stream, int n, const float4* pts, float h_area, float h_vol, KnnGrid* gp, int* cell_of,
                                   unsigned* cell_count, unsigned* cell_start, unsigned* cell_fill, float4* sorted) {
    int parts = (n + 2047) / 2048;
    parts = parts < 1 ? 1 : (parts > KNN_BOX_WGS ? KNN_BOX_WGS : parts);
    hipLaunchKernelGGL(knn_box_kernel, dim3(parts), dim3(256), 0, stream, n, pts, gp);
    hipLaunchKernelGGL(knn_grid_params_kernel, dim3(KNN_PARAM_WGS), dim3(1024), 0, stream, n, parts, h_area, h_vol, gp, cell_count);
    hipLaunchKernelGGL(knn_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, pts, gp, cell_of, cell_count);
    hipLaunchKernelGGL(knn_scan_sums_kernel, dim3(KNN_SCAN_TILES), dim3(1024), 0, stream, gp, cell_count);
    hipLaunchKernelGGL(knn_scan_kernel, dim3(KNN_SCAN_TILES), dim3(1024), 0, stream, gp, cell_count, cell_start, cell_fill);
    hipLaunchKernelGGL(knn_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, pts, cell_of, cell_fill, sorted);
}

// Tracker-sized clouds (<= KNN_FUSED_MAX_N points: 8-12 k per frame): the launches above as ONE single-workgroup kernel — bounding box,
// grid parameters, counting sort with the cell counters in LDS (global memory when the grid has more than KNN_LDS_CELLS cells), exclusive
// scan, fill.  Same parameters, same cell assignment, same cell_start table as the multi-launch path (the order of the points inside a
// cell is arbitrary in both; the search result does not depend on it).  Three kernel boundaries (~4 us each plus their launch gaps) less
// per tracked frame.
constexpr int KNN_FUSED_MAX_N = 32768;
constexpr int KNN_LDS_CELLS = 8192;
__global__ __launch_bounds__(1024) void knn_build_fused_kernel(int n, const float4* __restrict__ pts, float h_area, float h_vol,
                                                               KnnGrid* __restrict__ gp, int* __restrict__ cell_of,
                                                               unsigned* __restrict__ cell_count /* global fallback counters */,
                                                               unsigned* __restrict__ cell_start, unsigned* __restrict__ cell_fill,
                                                               float4* __restrict__ sorted) {
    __shared__ float s_lo[3][16], s_hi[3][16];
    __shared__ unsigned s_cnt[KNN_LDS_CELLS];
    __shared__ unsigned s_part[1024];
    __shared__ KnnGrid s_g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    // A depth frame's cloud (<= KNN_KEEP * 1024 points) stays in registers across the three passes: one trip to memory instead of three
    constexpr int KNN_KEEP = 12;
    const bool keep = n <= KNN_KEEP * 1024;
    float4 kept[KNN_KEEP];
    if (keep) {
#pragma unroll
        for (int u = 0; u < KNN_KEEP; ++u) { const int i = tid + u * 1024; kept[u] = pts[i < n ? i : 0]; }
#pragma unroll
        for (int u = 0; u < KNN_KEEP; ++u) {
            if (tid + u * 1024 < n) {
                const float4 p = kept[u];
                lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
                hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
            }
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < n; i += 1024) {   // (unrolled: the loads of several trips are in flight together — each pass is a chain of n / 1024 latencies otherwise)
            const float4 p = pts[i];
            lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
        }
        if (lane == 0) { s_lo[d][wave] = lo[d]; s_hi[d][wave] = hi[d]; }
    }
    __syncthreads();
    if (tid == 0) {
        float L[3], E[3];
        for (int d = 0; d < 3; ++d) {
            float l = s_lo[d][0], u = s_hi[d][0];
            for (int w = 1; w < 16; ++w) { l = fminf(l, s_lo[d][w]); u = fmaxf(u, s_hi[d][w]); }
            L[d] = l; E[d] = fmaxf(u - l, 0.f);
        }
        const KnnGrid g = knn_grid_from_box(L, E, n, h_area, h_vol);
        s_g = g;
        *gp = g;
    }
    __syncthreads();
    const KnnGrid g = s_g;
    const int ncells = g.ncells;
    const bool in_lds = ncells <= KNN_LDS_CELLS;
    unsigned* cnt = in_lds ? s_cnt : cell_count;
    for (int c = tid; c < ncells; c += 1024) cnt[c] = 0u;
    __syncthreads();
    int kept_cell[KNN_KEEP];
    if (keep) {
#pragma unroll
        for (int u = 0; u < KNN_KEEP; ++u) {
            kept_cell[u] = 0;
            if (tid + u * 1024 < n) {
                int cx, cy, cz;
                knn_cell_of(g, kept[u].x, kept[u].y, kept[u].z, cx, cy, cz);
                kept_cell[u] = (cz * g.ny + cy) * g.nx + cx;
                atomicAdd(&cnt[kept_cell[u]], 1u);
            }
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < n; i += 1024) {
            const float4 p = pts[i];
            int cx, cy, cz;
            knn_cell_of(g, p.x, p.y, p.z, cx, cy, cz);
            const int c = (cz * g.ny + cy) * g.nx + cx;
            cell_of[i] = c;
            atomicAdd(&cnt[c], 1u);
        }
    }
    __threadfence_block();
    __syncthreads();
    const int per = (ncells + 1023) / 1024;
    const int clo = tid * per, chi = (clo + per) < ncells ? (clo + per) : ncells;
    unsigned sum = 0;
    for (int c = clo; c < chi; ++c) sum += cnt[c];
    // exclusive scan of the 1024 partials: in-wave inclusive scan (six DPP / shuffle steps), then the 16 wave totals through LDS —
    // two barriers instead of the twenty of a Hillis-Steele scan over the whole workgroup
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    unsigned before = 0, total_all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const unsigned v = s_part[w]; if (w < wave) before += v; total_all += v; }
    unsigned run = before + incl - sum;
    unsigned* cur = in_lds ? s_cnt : cell_fill;     // the counters become the fill cursors
    for (int c = clo; c < chi; ++c) {
        const unsigned k = cnt[c];
        cell_start[c] = run;
        cur[c] = run;
        run += k;
    }
    if (tid == 1023) cell_start[ncells] = total_all;
    __threadfence_block();
    __syncthreads();
    if (keep) {
#pragma unroll
        for (int u = 0; u < KNN_KEEP; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                float4 p = kept[u];
                p.w = __int_as_float(i);
                sorted[atomicAdd(&cur[kept_cell[u]], 1u)] = p;
            }
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < n; i += 1024) {
            float4 p = pts[i];
            p.w = __int_as_float(i);
            sorted[atomicAdd(&cur[cell_of[i]], 1u)] = p;
        }
    }
}

typedef unsigned gi_uint2_t __attribute__((ext_vector_type(2)));
struct TopK { float my_d; int my_i; float tau_d; int tau_i; };
// (d, id) pairs are totally ordered (ids are unique; idle lanes carry (FLT_MAX, INT_MAX), the largest pair).  For d >= 0 (squared distances;
// candidates that are not < FLT_MAX — NaN, overflow — are turned into the idle pair where they are made) the order of the pairs IS the order
// of the 64-bit integers (bits of d) << 32 | id: one v_cmp_lt_u64 instead of three compares and two mask operations.
typedef unsigned long long pair_key_t;
__device__ __forceinline__ pair_key_t pair_key(float d, int i) { return ((pair_key_t)__float_as_uint(d) << 32) | (pair_key_t)(unsigned)i; }
__device__ __forceinline__ float pair_d(pair_key_t k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int pair_i(pair_key_t k) { return (int)(unsigned)k; }
__device__ __forceinline__ bool pair_less(float d, int i, float d2, int i2) { return pair_key(d, i) < pair_key(d2, i2); }
__device__ __forceinline__ void idle_unless_finite(float& d, int& id) { if (!(d < FLT_MAX)) { d = FLT_MAX; id = 0x7fffffff; } }

// value of lane (l ^ J) without the LDS crossbar: quad permutes (1, 2), two bank-masked row shifts (4), a row rotate (8), gfx950's
// v_permlane16_swap / v_permlane32_swap (16, 32).  Round 4b: the sorting network below used __shfl_xor = ds_bpermute_b32 + s_waitcnt, 42 LDS
// round trips per sort, and its compare-exchanges compiled into exec-masked branches (355 ds_bpermute / 243 s_and_saveexec in the kernel).
template <int J>
__device__ __forceinline__ unsigned lane_xor(unsigned v, int lane) {
    const int iv = (int)v;
    // (old = 0 with bound_ctrl: every lane has a valid source in these patterns, and the compiler then needs no copy of `old`)
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, iv, 0x12C, 0xF, 0xF, true);                            // row_ror 12: lane l <- (l + 4) mod 16, right for banks 0, 2
        return (unsigned)__builtin_amdgcn_update_dpp(t, iv, 0x124, 0xF, 0xA, false);                        // banks 1, 3: row_ror 4, lane l <- l - 4
    } else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, true);   // row_ror 8
    else if constexpr (J == 15) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x140, 0xF, 0xF, true);    // row_mirror
    else if constexpr (J == 16) {
        const gi_uint2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);    // x: rows (0, 0, 2, 2) of v, y: rows (1, 1, 3, 3)
        return __builtin_amdgcn_inverse_ballot_w64(0xFFFF0000FFFF0000ull) ? r.x : r.y;
    } else {
        static_assert(J == 32, "lane_xor: unsupported pattern");
        const gi_uint2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // x: lower half of v twice, y: upper half twice
        return __builtin_amdgcn_inverse_ballot_w64(0xFFFFFFFF00000000ull) ? r.x : r.y;
    }
}
template <int J>
__device__ __forceinline__ pair_key_t lane_xor_key(pair_key_t k, int lane) {
    return (pair_key_t)lane_xor<J>((unsigned)(k >> 32), lane) << 32 | (pair_key_t)lane_xor<J>((unsigned)k, lane);
}
// one compare-exchange of a bitonic network: partner = lane ^ J; blocks of K lanes alternate direction (K = 64: ascending everywhere)
// lanes that keep the smaller key in the step (K, J): a compile-time wave mask, handed to the selects as a scalar constant
// (computed from the lane id, the 21 masks of a sort cost ~120 vector instructions per wave and more scalar registers than there are)
template <int K, int J>
constexpr unsigned long long keep_min_mask() {
    unsigned long long m = 0ull;
    for (int l = 0; l < 64; ++l) {
        const bool up = K >= 64 || (l & K) == 0;
        if (((l & J) == 0) == up) m |= 1ull << l;
    }
    return m;
}
template <int K, int J>
__device__ __forceinline__ void cx_step(pair_key_t& k, int lane) {
    const bool keep_min = __builtin_amdgcn_inverse_ballot_w64(keep_min_mask<K, J>());
    if constexpr (J >= 16) {
        // the swap instructions hand BOTH partners to both lanes of a pair (a: the key of the lane with bit J clear, b: of the lane with it
        // set), so the step is a min / max of (a, b) — no second copy of the key, no select to rebuild "the other lane's key"
        gi_uint2_t hi, lo;
        if constexpr (J == 16) { hi = __builtin_amdgcn_permlane16_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
                                 lo = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false); }
        else { hi = __builtin_amdgcn_permlane32_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
               lo = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false); }
        const pair_key_t a = (pair_key_t)hi.x << 32 | lo.x, b = (pair_key_t)hi.y << 32 | lo.y;
        k = ((a < b) == keep_min) ? a : b;
    } else {
        const pair_key_t o = lane_xor_key<J>(k, lane);
        k = ((o < k) == keep_min) ? o : k;
    }
}
template <int K, int J>
__device__ __forceinline__ void cx_steps(pair_key_t& k, int lane) {     // J, J / 2, .. 1 at block size K
    cx_step<K, J>(k, lane);
    if constexpr (J > 1) cx_steps<K, J / 2>(k, lane);
}
// ascending bitonic sort of one pair per lane across the wave: 21 compare-exchange steps, no scalar round trips, no LDS
__device__ __forceinline__ void wave_sort_keys(pair_key_t& k, int lane) {
    cx_steps<2, 1>(k, lane); cx_steps<4, 2>(k, lane); cx_steps<8, 4>(k, lane);
    cx_steps<16, 8>(k, lane); cx_steps<32, 16>(k, lane); cx_steps<64, 32>(k, lane);
}
// the list (ascending over the lanes) and 64 candidates -> the 64 smallest of both, ascending: sort the candidates, take the lane-wise
// minimum against the reversed list (a bitonic sequence holding exactly the 64 smallest), finish with the six merge steps
__device__ inline void topk_merge64(TopK& t, float d, int id, const int kk, const int lane) {
    pair_key_t c = pair_key(d, id);
    wave_sort_keys(c, lane);
    if (__builtin_amdgcn_readfirstlane(t.my_i) == 0x7fffffff) {     // the list is still empty (ascending: lane 0 idle = every lane idle): the
        t.my_d = pair_d(c); t.my_i = pair_i(c);                      // sorted candidates ARE the list — the first batch of every query
    } else {
        c = lane_xor_key<15>(lane_xor_key<16>(lane_xor_key<32>(c, lane), lane), lane);     // lane 63 - l
        pair_key_t m = pair_key(t.my_d, t.my_i);
        m = c < m ? c : m;
        cx_steps<64, 32>(m, lane);
        t.my_d = pair_d(m); t.my_i = pair_i(m);
    }
    t.tau_d = readlane_f(t.my_d, kk - 1);
    t.tau_i = __builtin_amdgcn_readlane(t.my_i, kk - 1);
}
// Offer one candidate per lane (d = FLT_MAX / id = INT_MAX for idle lanes) to the cross-lane sorted list (ascending over the lanes; lanes
// >= kk hold the next larger pairs or the idle pair and are never read).  Few candidates below the current k-th pair are inserted one by one
// (a scalar round trip each, ~25 instructions); many — the first batches of a query, while the list is still filling — go through the
// sorting network instead (~200 instructions whatever their number).
constexpr int TOPK_SORT_FROM = 8;
__device__ inline void topk_offer(TopK& t, float d, int id, int kk, unsigned long long kmask, int lane) {
    idle_unless_finite(d, id);
    if (kk == 1) {      // nearest neighbour only (the exact-distance export): a lexicographic wave minimum, no list to maintain
        pair_key_t k = pair_key(d, id);
        { const pair_key_t o = lane_xor_key<32>(k, lane); k = o < k ? o : k; }
        { const pair_key_t o = lane_xor_key<16>(k, lane); k = o < k ? o : k; }
        { const pair_key_t o = lane_xor_key<8>(k, lane); k = o < k ? o : k; }
        { const pair_key_t o = lane_xor_key<4>(k, lane); k = o < k ? o : k; }
        { const pair_key_t o = lane_xor_key<2>(k, lane); k = o < k ? o : k; }
        { const pair_key_t o = lane_xor_key<1>(k, lane); k = o < k ? o : k; }
        if (k < pair_key(t.tau_d, t.tau_i)) { t.tau_d = t.my_d = pair_d(k); t.tau_i = t.my_i = pair_i(k); }   // every lane holds the same pair
        return;
    }
    unsigned long long m = __ballot(pair_less(d, id, t.tau_d, t.tau_i));
    if (__popcll(m) >= TOPK_SORT_FROM) { topk_merge64(t, d, id, kk, lane); return; }
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float cd = readlane_f(d, b);
        const int ci = __builtin_amdgcn_readlane(id, b);
        if (!pair_less(cd, ci, t.tau_d, t.tau_i)) continue;   // the list moved on since the ballot
        const unsigned long long le = __ballot(!pair_less(cd, ci, t.my_d, t.my_i)) & kmask;   // entries ranked before the candidate
        const int pos = __popcll(le);
        const float up_d = __int_as_float(wave_shr1(__float_as_int(t.my_d)));
        const int up_i = wave_shr1(t.my_i);
        if (lane > pos) { t.my_d = up_d; t.my_i = up_i; }
        else if (lane == pos) { t.my_d = cd; t.my_i = ci; }
        t.tau_d = readlane_f(t.my_d, kk - 1);
        t.tau_i = __builtin_amdgcn_readlane(t.my_i, kk - 1);
    }
}

// test hook (gsicp_debug_wave_sort): the wave-wide sorting network and the lane-exchange patterns it is made of, on one wave
__global__ __launch_bounds__(64) void debug_wave_sort_kernel(const float* __restrict__ d, const int* __restrict__ id, float* __restrict__ od,
                                                             int* __restrict__ oi, int* __restrict__ ox) {
    const int lane = threadIdx.x;
    pair_key_t k = pair_key(d[lane], id[lane]);
    wave_sort_keys(k, lane);
    od[lane] = pair_d(k); oi[lane] = pair_i(k);
    const unsigned me = (unsigned)lane * 3u + 1u;      // any lane-dependent value: ox[j][lane] must equal the value of lane (lane ^ J)
    ox[0 * 64 + lane] = (int)lane_xor<1>(me, lane); ox[1 * 64 + lane] = (int)lane_xor<2>(me, lane); ox[2 * 64 + lane] = (int)lane_xor<4>(me, lane);
    ox[3 * 64 + lane] = (int)lane_xor<8>(me, lane); ox[4 * 64 + lane] = (int)lane_xor<15>(me, lane); ox[5 * 64 + lane] = (int)lane_xor<16>(me, lane);
    ox[6 * 64 + lane] = (int)lane_xor<32>(me, lane);
}

// scan one contiguous range of the cell-sorted array, four batches per trip (four independent loads in flight: the whole-cloud fallback
// of a query in an empty neighbourhood is a chain of n / 64 batches, and with one load at a time each of them waited a memory latency)
__device__ inline void knn_scan_range(TopK& t, const float4& Q, const float4* __restrict__ sorted, unsigned s0, unsigned e0, int kk,
                                      unsigned long long kmask, int lane) {
    for (unsigned jb = s0; jb < e0; jb += 256) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned j = jb + 64u * u + lane;
            p[u] = sorted[j < e0 ? j : e0 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned j = jb + 64u * u + lane;
            if (jb + 64u * u >= e0) break;        // wave-uniform
            const bool in = j < e0;
            const float d = in ? dist2(Q.x, Q.y, Q.z, p[u].x, p[u].y, p[u].z) : FLT_MAX;
            const int id = in ? __float_as_int(p[u].w) : 0x7fffffff;
            topk_offer(t, d, id, kk, kmask, lane);
        }
    }
}
// Up to 64 runs of the cell-sorted array, one per lane (run = [cs, cs + cnt); cnt = 0 for lanes without one): their points are gathered 64
// per batch — lane l takes the l-th point of the concatenated runs, found by a binary search over the wave-resident prefix sums — so many
// short runs (a handful of points per cell) cost a few FULL batches instead of one sparse batch each.
__device__ inline void knn_gather_runs(TopK& t, const float4& Q, const float4* __restrict__ sorted, unsigned cs, unsigned cnt, int kk,
                                       unsigned long long kmask, int lane) {
    if (__ballot(cnt != 0u) == 0ull) return;
    unsigned incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += v;
    }
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    for (unsigned b0 = 0; b0 < total; b0 += 64) {
        const unsigned gi = b0 + (unsigned)lane;
        int lo = 0;                                   // number of runs that end at or before gi = the run holding gi
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const unsigned v = (unsigned)__shfl((int)incl, (lo + step - 1) & 63, 64);
            if (v <= gi) lo += step;
        }
        const int r = lo & 63;
        const unsigned e1 = (unsigned)__shfl((int)incl, r, 64), n1 = (unsigned)__shfl((int)cnt, r, 64), c1 = (unsigned)__shfl((int)cs, r, 64);
        float d = FLT_MAX;
        int id = 0x7fffffff;
        if (gi < total) {
            const float4 p = sorted[c1 + (gi - (e1 - n1))];
            d = dist2(Q.x, Q.y, Q.z, p.x, p.y, p.z);
            id = __float_as_int(p.w);
        }
        topk_offer(t, d, id, kk, kmask, lane);
    }
}
// conservative distance from coordinate q to the slab of cells [c0, c1] along one axis (0 inside); `slack` absorbs the
// rounding of the cell assignment, so a pruned range can never hold a point closer than the bound
__device__ inline float knn_gap(float q, float o, float h, int c0, int c1, float slack) {
    const float lo = o + (float)c0 * h, hi = o + (float)(c1 + 1) * h;
    const float g = q < lo ? lo - q : (q > hi ? q - hi : 0.f);
    return fmaxf(g - slack, 0.f);
}

// One WAVE per query.  Ring 1: the 27 cells around the query as nine runs (three x-adjacent cells each).  Ring r >= 2: the shell of the
// (2r+1)^3 cube as rows of the (2r+1)^2 (y, z) window — rows outside the previous window contribute their whole x segment, rows inside it
// their two end cells — each run pruned against the running k-th distance before it is gathered.  After each ring the k-th distance is
// tested against the distance to the faces of the scanned cube; a query still open after ring KNN_RMAX scans the whole cloud.
// Exact for any cell size.  The search is shared by the k-NN covariance pass, distCUDA2 and the exact 1-NN export: fills `t` with the kk
// nearest points of the cell-sorted cloud to Q; returns how the query settled (0 whole grid, 1 ring 1, 2 ring 2, 3 a later ring, 4 full scan).
constexpr int KNN_RMAX = 6;
__device__ inline int knn_search(const float4 Q, int kk, unsigned long long kmask, const KnnGrid& g, const unsigned* __restrict__ cell_start,
                                 const float4* __restrict__ sorted, int n, int lane, TopK& t) {
    const float slack = 2e-3f * g.h;
    int cx, cy, cz;
    knn_cell_of(g, Q.x, Q.y, Q.z, cx, cy, cz);
    t.my_d = FLT_MAX; t.my_i = 0x7fffffff; t.tau_d = FLT_MAX; t.tau_i = 0x7fffffff;
    int how = 4;
    bool done = false;
    for (int r = 1; r <= KNN_RMAX && !done; ++r) {
        const int x0 = cx - r < 0 ? 0 : cx - r, x1 = cx + r >= g.nx ? g.nx - 1 : cx + r;
        const int y0 = cy - r < 0 ? 0 : cy - r, y1 = cy + r >= g.ny ? g.ny - 1 : cy + r;
        const int z0 = cz - r < 0 ? 0 : cz - r, z1 = cz + r >= g.nz ? g.nz - 1 : cz + r;
        const int side = 2 * r + 1, rows = side * side;
        const float inv_side = 1.0f / (float)side;     // w / side below as a float product: exact for w < 169, side <= 13 ((w + 0.5) / side is never within 0.03 of an integer)
        // pass 0: every row of the window (outer rows: the whole segment; inner rows of r >= 2: the left end cell, x = cx - r);
        // pass 1 (r >= 2): the right end cell (x = cx + r) of the inner rows
        for (int pass = 0; pass < (r >= 2 ? 2 : 1); ++pass) {
            for (int w0 = 0; w0 < rows; w0 += 64) {
                const int w = w0 + lane;
                unsigned cs = 0, cnt = 0;
                if (w < rows) {
                    const int wz = (int)(((float)w + 0.5f) * inv_side);
                    const int dy = (w - wz * side) - r, dz = wz - r;
                    const int y = cy + dy, z = cz + dz;
                    const bool inner = r >= 2 && dy > -r && dy < r && dz > -r && dz < r;
                    if (y >= 0 && y < g.ny && z >= 0 && z < g.nz && (pass == 0 || inner)) {
                        int xa = x0, xb = x1;
                        if (inner) { xa = xb = (pass == 0 ? cx - r : cx + r); }
                        if (xa >= 0 && xb < g.nx) {
                            const int base = (z * g.ny + y) * g.nx;
                            cs = cell_start[base + xa];
                            cnt = cell_start[base + xb + 1] - cs;
                            if (cnt != 0u && r >= 2) {     // prune against the list of the previous rings (strictly farther runs only)
                                const float gx_ = knn_gap(Q.x, g.ox, g.h, xa, xb, slack), gy_ = knn_gap(Q.y, g.oy, g.h, y, y, slack),
                                            gz_ = knn_gap(Q.z, g.oz, g.h, z, z, slack);
                                if (gx_ * gx_ + gy_ * gy_ + gz_ * gz_ > t.tau_d) cnt = 0u;
                            }
                        }
                    }
                }
                knn_gather_runs(t, Q, sorted, cs, cnt, kk, kmask, lane);
            }
        }
        // every point outside the scanned cube is at least `gr` away (faces clipped by the grid boundary do not count: nothing lies beyond)
        float gr = FLT_MAX;
        if (x0 > 0) gr = fminf(gr, Q.x - (g.ox + (float)x0 * g.h));
        if (x1 < g.nx - 1) gr = fminf(gr, (g.ox + (float)(x1 + 1) * g.h) - Q.x);
        if (y0 > 0) gr = fminf(gr, Q.y - (g.oy + (float)y0 * g.h));
        if (y1 < g.ny - 1) gr = fminf(gr, (g.oy + (float)(y1 + 1) * g.h) - Q.y);
        if (z0 > 0) gr = fminf(gr, Q.z - (g.oz + (float)z0 * g.h));
        if (z1 < g.nz - 1) gr = fminf(gr, (g.oz + (float)(z1 + 1) * g.h) - Q.z);
        if (gr == FLT_MAX) { done = true; how = 0; }                      // the cube already covers the whole grid
        else {
            gr -= slack;                                                  // cell assignment is a rounded float multiply: stay conservative
            if (gr > 0.f && t.tau_d < gr * gr * 0.9999f) { done = true; how = r < 3 ? r : 3; }   // strict: an equal-distance outsider could win the index tie-break
        }
    }
    if (!done) {   // sparse neighbourhood: scan everything
        t.my_d = FLT_MAX; t.my_i = 0x7fffffff; t.tau_d = FLT_MAX; t.tau_i = 0x7fffffff;
        knn_scan_range(t, Q, sorted, 0u, (unsigned)n, kk, kmask, lane);
    }
    return how;
}

__global__ __launch_bounds__(256) void knn_grid_kernel(int n, int k, const float4* __restrict__ pts, const KnnGrid* __restrict__ gp,
                                                       const unsigned* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                       int* __restrict__ nbr_idx, float* __restrict__ nbr_d2, unsigned* __restrict__ ring_hist) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;                       // wave-uniform
    const int kk = k < n ? k : n;             // <= 64
    const unsigned long long kmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
    const KnnGrid g = *gp;
    TopK t;
    const int how = knn_search(pts[q], kk, kmask, g, cell_start, sorted, n, lane, t);
    if (lane < kk) {
        nbr_idx[(size_t)q * 64 + lane] = t.my_i;
        nbr_d2[(size_t)q * 64 + lane] = t.my_d;
    }
    if (lane == 0 && ring_hist) atomicAdd(&ring_hist[how], 1u);
}

// simple_knn._C.distCUDA2 [REF scene/gaussian_model.py:20]: mean squared distance to the 3 nearest OTHER points — the k = 4 instance of the
// same exact grid search (the query itself is one of the four; the entry carrying its own index is dropped, so coincident points count as
// neighbours at distance 0, as in the brute-force definition).  Replaces an O(P^2) LDS-tiled scan (ms at P = 300 k).
__global__ __launch_bounds__(256) void knn3_pack_kernel(int n, const float* __restrict__ xyz, float4* __restrict__ pts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pts[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f);
}
__global__ __launch_bounds__(256) void knn3_grid_kernel(int n, const float4* __restrict__ pts, const KnnGrid* __restrict__ gp,
                                                        const unsigned* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;                       // wave-uniform
    const int kk = n < 4 ? n : 4;
    const KnnGrid g = *gp;
    TopK t;
    (void)knn_search(pts[q], kk, (1ull << kk) - 1ull, g, cell_start, sorted, n, lane, t);
    // lanes 0..kk-1 hold the kk nearest in (d, index) order; add the (up to) three that are not the query, nearest first
    const unsigned long long others = __ballot(lane < kk && t.my_i != q);
    float s = 0.f;
    int taken = 0;
    for (unsigned long long m = others; m && taken < 3; m &= m - 1, ++taken) s += readlane_f(t.my_d, __ffsll((long long)m) - 1);
    if (lane == 0) out[q] = taken ? s / 3.0f : 0.0f;
}

// Exact nearest-target distance for the source points the gated search left without a neighbour (the reference exports the raw
// kd-tree distance whatever the gate): the same grid search with k = 1 over the TARGET's cell-sorted copy, instead of a scan of all
// targets per query.  Same dist2 arithmetic, and a minimum does not depend on the visiting order: bit-identical distances.
__global__ __launch_bounds__(256) void nn1_grid_kernel(const int* __restrict__ miss, const int* __restrict__ n_miss_p, const int* __restrict__ src_track,
                                                       const float4* __restrict__ src_pts, const double* __restrict__ lin_pose,
                                                       const KnnGrid* __restrict__ gp, const unsigned* __restrict__ cell_start,
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
    float4 Q;
    Q.x = ((Rf[0] * p.x + Rf[1] * p.y) + Rf[2] * p.z) + tf[0];
    Q.y = ((Rf[3] * p.x + Rf[4] * p.y) + Rf[5] * p.z) + tf[1];
    Q.z = ((Rf[6] * p.x + Rf[7] * p.y) + Rf[8] * p.z) + tf[2];
    Q.w = 0.f;
    const KnnGrid g = *gp;
    TopK t;
    (void)knn_search(Q, 1, 1ull, g, cell_start, sorted, n_tgt, lane, t);
    if (lane == 0) sqd[s] = t.my_d;
}

// One THREAD per point: mean / covariance of its neighbours in fp64 (summed in rank order, as the oracle does), cyclic
// Jacobi, quaternion (x,y,z,w), scales = sqrt(eigenvalues of the RAW covariance), regularised covariance for the cost.
__global__ __launch_bounds__(64) void cov_eig_kernel(int n, int k, const float4* __restrict__ pts, const int* __restrict__ nbr_idx,
                                                     const float* __restrict__ nbr_d2, float max_d2, int reg_method, int scale_mode,
                                                     double* __restrict__ cov, float* __restrict__ rotq, float* __restrict__ scales) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= n) return;
    const int kk = k < n ? k : n;
    double mu[3] = {0, 0, 0};
    double raw[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    if (kk <= 32) {
        // k <= 32 (the reference uses 20): the whole neighbour row lives in registers.  Distances, indices and then ALL point gathers are issued
        // side by side — three memory latencies for the thread instead of one or two per group of four neighbours, twice over (this kernel is
        // one wave per SIMD: its duration IS the thread's chain of dependent latencies).  Sums in rank order: bit-identical to the general path.
        const float* drow = nbr_d2 + (size_t)q * 64;
        const int* irow = nbr_idx + (size_t)q * 64;
        float d[32];
        int id[32];
        float4 P[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { d[u] = u < kk ? drow[u] : FLT_MAX; id[u] = irow[u < kk ? u : 0]; }
#pragma unroll
        for (int u = 0; u < 32; ++u) P[u] = pts[id[u]];
#pragma unroll
        for (int u = 0; u < 32; ++u) if (u < kk && u == cnt && d[u] <= max_d2) ++cnt;   // stops counting at the first neighbour beyond the radius
#pragma unroll
        for (int u = 0; u < 32; ++u) if (u < cnt) { mu[0] += (double)P[u].x; mu[1] += (double)P[u].y; mu[2] += (double)P[u].z; }
        mu[0] /= cnt; mu[1] /= cnt; mu[2] /= cnt;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (u < cnt) {
                const double dx = (double)P[u].x - mu[0], dy = (double)P[u].y - mu[1], dz = (double)P[u].z - mu[2];
                raw[0] += dx * dx; raw[1] += dx * dy; raw[2] += dx * dz; raw[3] += dy * dy; raw[4] += dy * dz; raw[5] += dz * dz;
            }
        }
    } else {
    // how many of the (distance-sorted) neighbours lie inside the k-NN radius: the row is scanned with independent loads first, so that the
    // point gathers below are not serialised behind a data-dependent exit
    {
        const float* drow = nbr_d2 + (size_t)q * 64;
        for (int j = 0; j < kk; j += 4) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = (j + u < kk) ? drow[j + u] : FLT_MAX;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j + u == cnt && d[u] <= max_d2) ++cnt;   // stops counting at the first neighbour beyond the radius
        }
    }
    const int* irow = nbr_idx + (size_t)q * 64;
    // four gathers in flight per trip; the additions keep the rank order (bit-identical to the one-at-a-time loop and to the oracle)
    for (int j = 0; j < cnt; j += 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[irow[j + u < cnt ? j + u : j]];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (j + u < cnt) { mu[0] += (double)p[u].x; mu[1] += (double)p[u].y; mu[2] += (double)p[u].z; }
    }
    mu[0] /= cnt; mu[1] /= cnt; mu[2] /= cnt;
    for (int j = 0; j < cnt; j += 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[irow[j + u < cnt ? j + u : j]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j + u < cnt) {
                const double dx = (double)p[u].x - mu[0], dy = (double)p[u].y - mu[1], dz = (double)p[u].z - mu[2];
                raw[0] += dx * dx; raw[1] += dx * dy; raw[2] += dx * dz; raw[3] += dy * dy; raw[4] += dy * dz; raw[5] += dz * dz;
            }
        }
    }
    }   // general path (k > 32)
#pragma unroll
    for (int d = 0; d < 6; ++d) raw[d] /= cnt;
    double ev[3], V[9], qd[4], out6[6];
    eig_sym3(raw, ev, V);
    rot_to_quat(V, qd);
    regularise(reg_method, ev, V, raw, out6);
#pragma unroll
    for (int d = 0; d < 4; ++d) rotq[4 * (size_t)q + d] = (float)qd[d];
#pragma unroll
    for (int d = 0; d < 3; ++d) scales[3 * (size_t)q + d] = (float)(scale_mode ? fmax(ev[d], 0.0) : sqrt(fmax(ev[d], 0.0)));   // variance | std-dev
#pragma unroll
    for (int d = 0; d < 6; ++d) cov[6 * (size_t)q + d] = out6[d];
}

__global__ __launch_bounds__(256) void cov_fromqs_kernel(int n, int reg_method, int scale_mode, const float* __restrict__ rots,
                                                         const float* __restrict__ scales, double* __restrict__ cov) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double q[4] = {rots[4 * (size_t)i], rots[4 * (size_t)i + 1], rots[4 * (size_t)i + 2], rots[4 * (size_t)i + 3]};
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nrm > 0) { q[0] /= nrm; q[1] /= nrm; q[2] /= nrm; q[3] /= nrm; } else { q[0] = q[1] = q[2] = 0; q[3] = 1; }
    double R[9];
    quat_to_rot(q, R);
    const double s0 = scales[3 * (size_t)i], s1 = scales[3 * (size_t)i + 1], s2 = scales[3 * (size_t)i + 2];
    const double v[3] = {scale_mode ? s0 : s0 * s0, scale_mode ? s1 : s1 * s1, scale_mode ? s2 : s2 * s2};   // scales are variances | std-devs
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
// sort key = coarse-cell key << 3 | octant: a coarse cell's points come out contiguous, ordered by fine cell, in track order within one
__global__ __launch_bounds__(256) void grid_keys_kernel(int n_track, const int* __restrict__ track, const float4* __restrict__ pts,
                                                        float inv_hf, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_track) return;
    const int i = track[s];
    const float4 p = pts[i];
    const int fx = (int)floorf(p.x * inv_hf), fy = (int)floorf(p.y * inv_hf), fz = (int)floorf(p.z * inv_hf);
    keys[s] = cell_key(fx >> 1, fy >> 1, fz >> 1) << 3 | (unsigned long long)((fx & 1) | (fy & 1) << 1 | (fz & 1) << 2);
    vals[s] = (unsigned)i;
}
// Occupied coarse cells = runs of the sorted keys.  Cell ordinals are the RANKS of the run heads (rounds 3-4a handed them out with one
// same-address atomic per cell: 17 k serialised atomics = 150 us of a 1e6-point build): heads per slab of 256 or 2 048 sorted keys, one
// single-workgroup scan over the slabs (its total is what the host reads back to size the table by CELLS), rank inside the slab by ballots.
// sorted keys per workgroup = 256 x NJ (element j * 256 + t of the slab for thread t): NJ = 8 for maps, 1 below GRID_SMALL_N points (a frame-sized
// target would otherwise be five workgroups walking eight insertions each)
constexpr int GRID_SMALL_N = 65536;
__device__ inline bool grid_run_head(const unsigned long long* __restrict__ skeys, int s, int n) {
    return s < n && (s == 0 || (skeys[s - 1] >> 3) != (skeys[s] >> 3));
}
template <int NJ>
__global__ __launch_bounds__(256) void grid_heads_kernel(int n, const unsigned long long* __restrict__ skeys, unsigned* __restrict__ slab_heads) {
    __shared__ unsigned s_w[4];
    const int tid = threadIdx.x, base = (int)blockIdx.x * (256 * NJ);
    unsigned c = 0u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) c += (unsigned)__popcll(__ballot(grid_run_head(skeys, base + j * 256 + tid, n)));
    if ((tid & 63) == 0) s_w[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) slab_heads[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
// exclusive scan of the slab counts in place (one workgroup; <= 2^26 / 2048 = 32 768 slabs); n_cells[0] = total = occupied cells
__global__ __launch_bounds__(1024) void grid_heads_scan_kernel(int n_slabs, unsigned* __restrict__ slab_heads, unsigned* __restrict__ n_cells) {
    __shared__ unsigned s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_slabs + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < n_slabs ? lo + per : n_slabs;
    unsigned sum = 0u;
    for (int b = lo; b < hi; ++b) sum += slab_heads[b];
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum, total = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const unsigned v = s_w[w]; if (w < wave) run += v; total += v; }
    for (int b = lo; b < hi; ++b) { const unsigned c = slab_heads[b]; slab_heads[b] = run; run += c; }
    if (tid == 0) n_cells[0] = total;
}
// sorted (key, original index) -> sorted float4 records; the first element of each COARSE run inserts {coarse key -> its rank} in the
// table and clears the cell's eight {begin, count} pairs
template <int NJ>
__global__ __launch_bounds__(256) void grid_cells_kernel(int n, const unsigned long long* __restrict__ skeys, const unsigned* __restrict__ svals,
                                                         const float4* __restrict__ pts, float4* __restrict__ sorted, unsigned mask,
                                                         unsigned long long* __restrict__ tkeys, unsigned* __restrict__ tords,
                                                         uint2* __restrict__ cells, const unsigned* __restrict__ slab_base) {
    __shared__ unsigned s_cnt[NJ][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, base = (int)blockIdx.x * (256 * NJ);
    unsigned id[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int s = base + j * 256 + tid; id[j] = svals[s < n ? s : n - 1]; }
    float4 p[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) p[j] = pts[id[j]];
    unsigned long long heads[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        heads[j] = __ballot(grid_run_head(skeys, base + j * 256 + tid, n));
        if (lane == 0) s_cnt[j][wave] = (unsigned)__popcll(heads[j]);
    }
    __syncthreads();
    unsigned before = slab_base[blockIdx.x];     // heads in front of (j, wave) in slab order
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int s = base + j * 256 + tid;
        unsigned mine = before;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned v = s_cnt[j][w]; if (w < wave) mine += v; before += v; }
        if (s >= n) continue;
        float4 q = p[j];
        q.w = __int_as_float((int)id[j]);
        sorted[s] = q;
        if (!(heads[j] >> lane & 1ull)) continue;
        const unsigned ord = mine + (unsigned)__popcll(heads[j] & ((1ull << lane) - 1ull));
        const unsigned long long key = skeys[s] >> 3;
        unsigned slot = (hash_key(key) << 2) & mask;   // first slot of the key's 4-slot bucket (grid_nn reads whole buckets)
        for (;;) {
            const unsigned long long prev = atomicCAS(&tkeys[slot], EMPTY_KEY, key);
            if (prev == EMPTY_KEY) break;
            slot = (slot + 1) & mask;
        }
        tords[slot] = ord;
        uint4* c4 = (uint4*)(cells + (size_t)ord * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) c4[k] = make_uint4(0u, 0u, 0u, 0u);
    }
}
// second pass (the table is complete): the first element of each FINE run looks its coarse cell up and writes {begin, count}
__global__ __launch_bounds__(256) void grid_octants_kernel(int n, const unsigned long long* __restrict__ skeys, unsigned mask,
                                                           const unsigned long long* __restrict__ tkeys, const unsigned* __restrict__ tords,
                                                           uint2* __restrict__ cells) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const unsigned long long fk = skeys[s];
    if (s != 0 && skeys[s - 1] == fk) return;
    unsigned cnt = 1;
    while (s + (int)cnt < n && skeys[s + cnt] == fk) ++cnt;
    const unsigned long long key = fk >> 3;
    unsigned slot = (hash_key(key) << 2) & mask;
    while (tkeys[slot] != key) slot = (slot + 1) & mask;      // present by construction
    cells[(size_t)tords[slot] * 8 + (unsigned)(fk & 7ull)] = make_uint2((unsigned)s, cnt);
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

// Page-locked, device-visible host memory the kernels write their small results into directly: the host polls the sequence
// numbers instead of paying for a D2H copy plus a stream synchronise (each 15-60 us of fixed cost against ~100 us kernels).
struct HostMailbox {
    AlignResult result;
    unsigned align_seq;          // written (system scope) after `result`
    unsigned export_seq;         // written after the correspondence export below
    unsigned scratch_u32;        // small read-backs (selection count)
};

constexpr int AL_T = 256;        // threads per workgroup
constexpr int AL_MAX_WG = 240;   // partial-sum table size; the launch is further bounded by the device's occupancy (gsicp_gicp_create) so that
                                 // every workgroup is resident and the grid barrier is safe; a barrier that still times out re-runs as 1 workgroup

struct AlignSync {               // zeroed once at creation; every launch leaves it zeroed again (the last workgroup out resets it)
    unsigned counter;
    unsigned abort;
    unsigned exit_count;
    unsigned pad[29];
    double partials[2][AL_MAX_WG][NRED + 1];   // + the trial cost that rides along with a speculative linearisation
};

struct AlignArgs {
    int n_src;                 // trackable source points
    const int* src_track;
    const float4* src_pts;
    const double* src_cov;
    const float4* tgt_pts;     // original order
    const double* tgt_cov;
    GridView grid;             // complete within the gate radius
    GridView fine;             // dense maps only (use_grid = 0 otherwise): the same structure at a smaller radius, tried first
    float fine_r2;             // squared radius the fine level is complete within
    float gate;                // squared correspondence gate (FLT_MAX = none)
    double init[12];           // R, t
    int max_iter, lm_max_iter;
    double rot_eps, trans_eps, lm_init;
    int* corr;                 // per trackable source point
    float* sqd;
    double* maha;              // 6 per trackable source point
    int* corr2;                // shadow set for the speculative linearisation (same sizes)
    float* sqd2;
    double* maha2;
    AlignResult* result;
    AlignSync* sync;
    int* miss_counter;         // zeroed here for the exact-distance pass that follows (saves a memset launch per frame)
    HostMailbox* mailbox;      // pinned host memory
    unsigned seq;              // this launch's sequence number
    unsigned long long* trace; // diagnostics (GSICP_ALIGN_TRACE): wall_clock64 stamps of workgroup 0 at phase boundaries, or NULL
    int wave_prio;             // round 6 (GSICP_TRACKER_WAVE_PRIO): s_setprio level of this kernel's waves — next to the mapper's throughput waves on the same SIMDs the
                               // tracker's 33 latency-critical workgroups win the issue arbitration (every grid-wide phase ends when the slowest workgroup arrives)
};

__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ inline bool solve6(const double* H, const double* b, double* x) {
    // fully unrolled so that L / Dg / y live in registers (with runtime loop bounds they went to scratch memory and this serial
    // step cost 4-5 us of every LM phase); the arithmetic and its order are unchanged
    double L[36], Dg[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = H[6 * j + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= L[6 * j + k] * L[6 * j + k] * Dg[k];
        Dg[j] = d;
        if (d == 0.0 || !isfinite(d)) ok = false;
        const double inv_d = 1.0 / d;   // one fp64 division (~40 instructions on a single lane) per column instead of one per entry
        L[6 * j + j] = 1.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = H[6 * i + j];
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k < j) v -= L[6 * i + k] * L[6 * j + k] * Dg[k];
                L[6 * i + j] = v * inv_d;
            }
        }
    }
    if (!ok) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < i) v -= L[6 * i + k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] /= Dg[i];
    double xs[6];
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) {
        const int i = 5 - ii;
        double v = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k > i) v -= L[6 * k + i] * xs[k];
        xs[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = xs[i];
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

__device__ inline double readlane_d(double v, int l) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The scalar steps between two grid-wide phases used to run on ONE lane while the rest of the chip waited (5.8 us of every ~19 us phase).
// The variants below are called by all 64 lanes of a workgroup's first wave with identical inputs: what is independent runs on
// different lanes — the four trigonometric calls of the exponential map become two (lane 1 evaluates the full angle while the
// others evaluate the half angle), the twelve fp64 divisions of the convergence test become one — and every value is the one the
// single-lane code computed (same operations on the same operands; max and the lane a value is computed on do not change it).
__device__ inline void se3_exp_wave(const double* a, double* R, double* t, const int lane) {
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real, theta = 0, s_th = 0, c_th = 1;
    bool have_full = false;
    if (theta_sq < 1e-10) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        const double arg = lane == 1 ? theta : half;
        const double sv = sin(arg), cv = cos(arg);
        imag = readlane_d(sv, 0) / theta;
        real = readlane_d(cv, 0);
        s_th = readlane_d(sv, 1); c_th = readlane_d(cv, 1);
        have_full = true;
    }
    const double q[4] = {imag * wx, imag * wy, imag * wz, real};
    quat_to_rot(q, R);
    double V[9];
    if (theta_sq < 1e-20) {
        for (int i = 0; i < 9; ++i) V[i] = R[i];
    } else {
        if (theta == 0) theta = sqrt(theta_sq);
        if (!have_full) { s_th = sin(theta); c_th = cos(theta); }
        const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
        const double c1 = (1.0 - c_th) / theta_sq, c2 = (theta - s_th) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) t[i] = V[3 * i] * a[3] + V[3 * i + 1] * a[4] + V[3 * i + 2] * a[5];
}
__device__ inline bool is_converged_wave(const double* R, const double* t, double rot_eps, double trans_eps, const int lane) {
    double v = 0.0;
    if (lane < 9) v = fabs(R[lane] - (lane % 4 == 0 ? 1.0 : 0.0)) / rot_eps;
    else if (lane < 12) v = fabs(t[lane - 9]) / trans_eps;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));     // lanes 0..15 hold the twelve terms (and zeros)
    return readlane_d(v, 0) < 1.0;
}
struct AlignShared {
    double scratch[AL_T / 64][NRED + 1];
    double chunk[8][32];   // partial sums of an eighth of the workgroups each (grid_sum)
    double red[NRED + 1];
    double x0[12];      // current pose R,t
    double xi[12];      // trial pose
    double delta[12];
    double H[36], b[6];
    double spec[NRED];  // H (21), b (6), cost (1) of the speculative linearisation at the accepted trial pose
    double y0, lambda, nu, denom;
    int state;          // 0 continue LM trials, 1 step accepted / done with this outer iteration, 2 abort
    int converged;
    int abort;
    int accepted;       // the last LM trial moved x0 to xi (so the speculative linearisation at xi is the next iteration's)
};

// Grid-wide sum of NV doubles per thread.  Workgroup partials go to global memory, one monotonic-counter grid barrier
// (agent-scope release before the arrive, relaxed polling by ONE lane, agent-scope acquire after — the protocol of
// cdna_hip_programming.md §6 G16), then every workgroup adds the partials in the same fixed order, so all workgroups
// hold bit-identical totals and take identical decisions without any further communication.
typedef unsigned gi_uint2 __attribute__((ext_vector_type(2)));
// (A_lo + A_hi | B_lo + B_hi) per 32-lane half: the sum over the two halves of value A lands in the lower half, of B in the upper
__device__ inline double fold_swap32(double a, double b) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    const gi_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    const gi_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    const double x = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    const double y = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
    return x + y;
}
// rows (A_r0 + A_r1, B_r0 + B_r1, A_r2 + A_r3, B_r2 + B_r3)
__device__ inline double fold_swap16(double a, double b) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    const gi_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const gi_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    const double x = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    const double y = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
    return x + y;
}
template <int CTRL>
__device__ inline double dpp_move_d(double v) {   // v as seen through a DPP lane pattern (invalid source lanes read 0: bound_ctrl)
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the 16 lanes of each row; valid in the row's last lane (row_shr 1, 2, 4, 8: lane i accumulates lanes i-15..i)
__device__ inline double row_sum_last(double v) {
    v += dpp_move_d<0x111>(v);   // row_shr:1
    v += dpp_move_d<0x112>(v);   // row_shr:2
    v += dpp_move_d<0x114>(v);   // row_shr:4
    v += dpp_move_d<0x118>(v);   // row_shr:8
    return v;
}

__device__ inline void trace_stamp(unsigned long long* trace, int& n, int tag) {
    if (trace && blockIdx.x == 0 && threadIdx.x == 0 && n < 250) { trace[2 * n] = (unsigned long long)tag; trace[2 * n + 1] = wall_clock64(); ++n; }
}
// The fence-free grid barrier inside grid_sum rests on gfx94x / gfx950 behaviour: agent-scope atomic stores are written through to
// the coherence point, vmcnt covers stores, and a thread re-reads after the barrier only what it wrote itself (the same s = gtid + k * gstride
// mapping in linearize_points, the trial-cost loop and the final copy).  This file is built for gfx950 only; refuse anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gicp.hip: the grid barrier of gicp_align_kernel assumes gfx942 / gfx950 memory behaviour (see grid_sum)"
#endif
template <int NV>
__device__ inline void grid_sum(double* vals, AlignShared& sh, AlignSync* sy, unsigned& epoch, int tid, unsigned long long* trace = nullptr,
                                int* trace_n = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x;
    // Wave reduction of NV doubles without the LDS crossbar: gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves /
    // odd-even rows of two registers, so two VALUES fold into one per level (NV -> NV/2 -> NV/4 values), and only the remaining
    // quarter goes through the four in-row DPP steps.  ~6 instructions per value instead of 18 ds_bpermute-based ones (8.5 us ->
    // ~1 us per phase).  Row r of folded value q holds original value 4q + {0, 2, 1, 3}[r]; the row total sits in the row's last lane.
    constexpr int N2 = (NV + 1) / 2, N4 = (N2 + 1) / 2;
    double f2[N2], f4[N4];
#pragma unroll
    for (int k = 0; k < N2; ++k) {
        const double a0 = vals[2 * k], b0 = (2 * k + 1 < NV) ? vals[2 * k + 1] : 0.0;
        f2[k] = fold_swap32(a0, b0);
    }
#pragma unroll
    for (int k = 0; k < N4; ++k) {
        const double a0 = f2[2 * k], b0 = (2 * k + 1 < N2) ? f2[2 * k + 1] : 0.0;
        f4[k] = fold_swap16(a0, b0);
    }
#pragma unroll
    for (int k = 0; k < N4; ++k) f4[k] = row_sum_last(f4[k]);
    if ((lane & 15) == 15) {
        const int r = lane >> 4;
        const int sub = r == 0 ? 0 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
#pragma unroll
        for (int k = 0; k < N4; ++k) {
            const int idx = 4 * k + sub;
            if (idx < NV) sh.scratch[wave][idx] = f4[k];
        }
    }
    __syncthreads();
    const int buf = epoch & 1;
    if (tid < NV) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < AL_T / 64; ++w) s += sh.scratch[w][tid];
        // agent-scope atomic store: written through to where every XCD reads it (see the note at the barrier)
        __hip_atomic_store(&sy->partials[buf][blockIdx.x][tid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ++epoch;
    if (trace) trace_stamp(trace, *trace_n, 10);   // wave + LDS reduction done, partial stored
    if (nwg > 1) {
        __syncthreads();
        if (tid == 0) {
            // NO agent-scope fences around this barrier (round 3).  The only data that crosses workgroups here are the partial sums, and they
            // travel as agent-scope atomic stores / loads (write-through, read at the coherence point); everything else a thread reads after the
            // barrier it wrote itself (its points' correspondences and Mahalanobis matrices) or was written before the launch (grid, clouds,
            // covariances).  A release FENCE would write back every dirty line of this XCD's L2 — the co-tenant mapper's included — and an
            // acquire FENCE would invalidate the L2, so that the hash-grid search after every barrier restarted from cold lines.
            // All partial stores of this workgroup were issued by this wave (NV <= 64): waiting for its own memory operations orders them
            // before the arrive.
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
            sh.abort = ab;
        }
    }
    __syncthreads();
    if (trace) trace_stamp(trace, *trace_n, 11);   // barrier passed
    // All 256 threads fetch: thread (k = tid & 31, c = tid >> 5) adds the partials of value k over the c-th eighth of the workgroups, in
    // workgroup order; then the eight chunk sums are added in chunk order.  ONE round of loads in flight instead of ceil(nwg / 8) dependent
    // rounds on 29 threads (1.6 -> ~0.9 us of every phase); the order is fixed, so every workgroup still holds the same bits.
    {
        const int k = tid & 31, c = tid >> 5, per = (nwg + 7) >> 3;
        double s = 0;
        if (k < NV) {
            const int w_lo = c * per, w_hi = (w_lo + per) < nwg ? (w_lo + per) : nwg;
            for (int w0 = w_lo; w0 < w_hi; w0 += 8) {     // one trip for up to 64 workgroups
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (w0 + u < w_hi) ? __hip_atomic_load(&sy->partials[buf][w0 + u][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) if (w0 + u < w_hi) s += v[u];
            }
            sh.chunk[c][k] = s;
        }
    }
    __syncthreads();
    if (tid < NV) {
        double s = sh.chunk[0][tid];
#pragma unroll
        for (int c = 1; c < 8; ++c) s += sh.chunk[c][tid];
        sh.red[tid] = s;
    }
    __syncthreads();
    if (trace) trace_stamp(trace, *trace_n, 12);   // partials of all workgroups summed
}

// Linearisation at pose x (R 9, t 3): nearest target inside the gate (fp32 search, fixed order), Mahalanobis matrix, and this
// thread's share of H (21), b (6), cost (1) over its strided source points.  Writes the correspondences / distances / matrices
// of the points it owns into the given buffer set.
__device__ __forceinline__ void linearize_points(const AlignArgs& a, const double* __restrict__ x, int* __restrict__ corr, float* __restrict__ sqd,
                                        double* __restrict__ maha, int gtid, int gstride, double* __restrict__ acc, uint2* __restrict__ queue, int* tn = nullptr) {
    // Register diet (round 3): the pose stays in LDS (`x` points at sh.x0 / sh.xi; every read is a broadcast) instead of 24 + 12 live
    // registers, and the 3x6 Jacobian J = [skew(T p) | -I] is never formed: its zeros and -1s are folded by hand.  Folding is EXACT, not an
    // approximation — 0 * finite = 0, 0 + v = v, (-1) * v = -v, u + (-v) = u - v in IEEE arithmetic (no contraction in this file) — and the
    // non-zero terms are added in the order the generic triple products J^T (M J), J^T (M e) add them, so H, b and the cost are the same bits
    // as before (and as the oracle's generic evaluation): 70 fp64 operations per point instead of ~250, 18 live temporaries instead of ~60.
    for (int s = gtid; s < a.n_src; s += gstride) {
        const int i = a.src_track[s];
        const float4 p = a.src_pts[i];
        float qx, qy, qz;
        {
            const float r0 = (float)x[0], r1 = (float)x[1], r2 = (float)x[2], r3 = (float)x[3], r4 = (float)x[4], r5 = (float)x[5];
            const float r6 = (float)x[6], r7 = (float)x[7], r8 = (float)x[8];
            qx = ((r0 * p.x + r1 * p.y) + r2 * p.z) + (float)x[9];
            qy = ((r3 * p.x + r4 * p.y) + r5 * p.z) + (float)x[10];
            qz = ((r6 * p.x + r7 * p.y) + r8 * p.z) + (float)x[11];
        }
        float bd; int bi;
        if (tn) trace_stamp(a.trace, *tn, 20);
        // A neighbour found inside the fine level's radius is THE nearest neighbour (that level is complete within its radius, and
        // anything nearer would lie inside it too); otherwise the gate-sized level answers.  Same (d, id) either way.
        // (One copy of the search code: a loop over the levels, not two inlined calls — the kernel must stay inlinable as a whole, or the
        // fp64 accumulators and the argument block travel through scratch memory.)
#pragma nounroll
        for (int lv = a.fine.use_grid ? 0 : 1; lv < 2; ++lv) {
            GridView gv = lv == 0 ? a.fine : a.grid;
            grid_nn(gv, qx, qy, qz, bd, bi, queue);
            if (lv == 0 && bd < a.fine_r2) break;
        }
        if (tn) trace_stamp(a.trace, *tn, 21);
        sqd[s] = bd;
        int c = -1;
        if (bi >= 0 && bd < a.gate) {
            double m[6];
            bool ok;
            {
                const double* A = a.src_cov + 6 * (size_t)i;
                const double* B = a.tgt_cov + 6 * (size_t)bi;
                const double Am[9] = {A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5]};
                double R[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = x[k];
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
                ok = inv_sym3(S, m);
            }
            if (ok) {
                c = bi;
#pragma unroll
                for (int d = 0; d < 6; ++d) maha[6 * (size_t)s + d] = m[d];
                const float4 bp = a.tgt_pts[bi];
                double ta[3], e[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) ta[r] = x[3 * r] * (double)p.x + x[3 * r + 1] * (double)p.y + x[3 * r + 2] * (double)p.z + x[9 + r];
                e[0] = (double)bp.x - ta[0]; e[1] = (double)bp.y - ta[1]; e[2] = (double)bp.z - ta[2];
                const double tx = ta[0], ty = ta[1], tz = ta[2];
                // M rows: (m0 m1 m2) (m1 m3 m4) (m2 m4 m5)
                double Me[3];
                Me[0] = m[0] * e[0] + m[1] * e[1] + m[2] * e[2];
                Me[1] = m[1] * e[0] + m[3] * e[1] + m[4] * e[2];
                Me[2] = m[2] * e[0] + m[4] * e[1] + m[5] * e[2];
                acc[27] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
                // MJ = M J, columns 0..2 (M skew(t)); columns 3..5 are -M.  Row r of M: (Mr0, Mr1, Mr2).
                //   MJ[r][0] = Mr1 tz - Mr2 ty,  MJ[r][1] = Mr2 tx - Mr0 tz,  MJ[r][2] = Mr0 ty - Mr1 tx
                const double M00 = m[0], M01 = m[1], M02 = m[2], M11 = m[3], M12 = m[4], M22 = m[5];
                const double q01 = M02 * tx - M00 * tz, q02 = M00 * ty - M01 * tx;      // (MJ[0][0] only feeds the lower triangle)
                const double q10 = M11 * tz - M12 * ty, q11 = M12 * tx - M01 * tz, q12 = M01 * ty - M11 * tx;
                const double q20 = M12 * tz - M22 * ty, q21 = M22 * tx - M02 * tz, q22 = M02 * ty - M12 * tx;
                // H = J^T (M J), upper triangle row by row (acc[0..20]); column r of J: (0, tz, -ty) (-tz, 0, tx) (ty, -tx, 0) then -I
                //   H[0][c] = tz MJ[1][c] - ty MJ[2][c];  H[1][c] = tx MJ[2][c] - tz MJ[0][c];  H[2][c] = ty MJ[0][c] - tx MJ[1][c];  H[3+i][c] = -MJ[i][c]
                acc[0] += tz * q10 - ty * q20;
                acc[1] += tz * q11 - ty * q21;
                acc[2] += tz * q12 - ty * q22;
                acc[3] += ty * M02 - tz * M01;          // tz (-M10) - ty (-M20)
                acc[4] += ty * M12 - tz * M11;
                acc[5] += ty * M22 - tz * M12;
                acc[6] += tx * q21 - tz * q01;
                acc[7] += tx * q22 - tz * q02;
                acc[8] += tz * M00 - tx * M02;          // tx (-M20) - tz (-M00)
                acc[9] += tz * M01 - tx * M12;
                acc[10] += tz * M02 - tx * M22;
                acc[11] += ty * q02 - tx * q12;
                acc[12] += tx * M01 - ty * M00;         // ty (-M00) - tx (-M10)
                acc[13] += tx * M11 - ty * M01;
                acc[14] += tx * M12 - ty * M02;
                acc[15] += M00; acc[16] += M01; acc[17] += M02;
                acc[18] += M11; acc[19] += M12;
                acc[20] += M22;
                // b = J^T (M e)
                acc[21] += tz * Me[1] - ty * Me[2];
                acc[22] += tx * Me[2] - tz * Me[0];
                acc[23] += ty * Me[0] - tx * Me[1];
                acc[24] += -Me[0];
                acc[25] += -Me[1];
                acc[26] += -Me[2];
            }
        }
        corr[s] = c;
        if (tn) trace_stamp(a.trace, *tn, 22);
    }
}

// The whole Levenberg-Marquardt loop in one persistent launch.  Every grid-wide phase costs ~15 us of cross-XCD barrier and
// reduction latency whatever it computes, so the phase count is what matters: the trial-cost phase also linearises at the trial
// pose into a shadow buffer set and reduces both in ONE barrier.  When the trial is accepted (the common case: every step of
// the benchmark pairs) the next outer iteration starts from that linearisation instead of spending a phase on it; when it is
// rejected the speculative result is dropped.  The arithmetic, its order and therefore every result bit are those of the
// phase-per-linearisation schedule (and of the oracle).
//
// Round 4: ONE phase site.  The loop below is a state machine whose every pass ends in the same code — [trial cost with the frozen
// correspondences, unless this is the opening linearisation] + linearise + grid-wide sum of 29 values — instead of two inlined copies
// of the linearisation (the opening one at x0, the speculative one at the trial pose).  The opening pass reduces a 29th value that is
// zero; values 0..27 go through the same folds and adds as before, so every bit is unchanged.  Half the code (the search alone is
// ~1 k instructions), which keeps the kernel inlined as a whole.
__global__ __launch_bounds__(AL_T) void gicp_align_kernel(AlignArgs a) {
    if (a.wave_prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (a.wave_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.wave_prio == 1) __builtin_amdgcn_s_setprio(1);
    __shared__ AlignShared sh;
    __shared__ uint2 s_queue[26 * AL_T];     // grid_nn's per-lane work list of non-empty neighbour cells: entry k of thread t at [k * AL_T + t]
    static_assert(AL_T == AL_T_CONST, "grid_nn's work-list stride");
    const int tid = threadIdx.x;
    const int gtid = blockIdx.x * AL_T + tid, gstride = gridDim.x * AL_T;
    const bool leader = blockIdx.x == 0 && tid == 0;
    unsigned epoch = 0;
    if (tid < 12) sh.x0[tid] = a.init[tid];
    if (tid == 0) { sh.lambda = -1.0; sh.converged = 0; sh.abort = 0; sh.accepted = 0; }
    if (leader && a.miss_counter) *a.miss_counter = 0;
    __syncthreads();

    int iterations = 0, lm_trials = 0, failed = 0;
    int tn = 0;
    trace_stamp(a.trace, tn, 0);
    int cur = 0;               // which buffer set holds the linearisation in use (wave-uniform, identical in every workgroup)
    bool opening = true;       // the next pass is a plain linearisation at x0 (the first pass, or after a step that left no speculative one)
    int it = 0, trial = 0;

    while (it < a.max_iter) {
        int* corr_c = cur ? a.corr2 : a.corr;
        double* maha_c = cur ? a.maha2 : a.maha;
        // where this pass's linearisation goes: the set in use (opening pass) or the shadow set (speculative, at the trial pose)
        int* corr_w = (cur != 0) == opening ? a.corr2 : a.corr;
        float* sqd_w = (cur != 0) == opening ? a.sqd2 : a.sqd;
        double* maha_w = (cur != 0) == opening ? a.maha2 : a.maha;
        if (!opening) {
            if (trial == 0) {
                // ---------------- start of an outer iteration: H, b, y0 from the linearisation in use
                if (tid < 64) {   // first wave, one element per lane
                    if (tid < 36) {
                        const int r = tid / 6, c = tid % 6, lo = r < c ? r : c, hi = r < c ? c : r;
                        sh.H[tid] = sh.spec[6 * lo - lo * (lo - 1) / 2 + (hi - lo)];       // spec holds the upper triangle row by row
                    }
                    if (tid < 6) sh.b[tid] = sh.spec[21 + tid];
                    if (tid == 0) {
                        sh.y0 = sh.spec[27];
                        if (sh.lambda < 0.0) {
                            const int diag[6] = {0, 6, 11, 15, 18, 20};
                            double mx = 0;
                            for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(sh.spec[diag[i]]));
                            sh.lambda = a.lm_init * mx;
                        }
                        sh.nu = 2.0;
                        sh.accepted = 0;
                    }
                    if (blockIdx.x == 0 && tid < 12) a.result->lin_pose[tid] = sh.x0[tid];
                }
                __syncthreads();
                trace_stamp(a.trace, tn, 30);
            }
            if (trial >= a.lm_max_iter) { failed = 1; break; }   // "lm not converged"
            // ---------------- LM trial (every workgroup runs the same scalar arithmetic on the same totals)
            ++lm_trials;
            if (tid < 64) {   // every lane of the first wave runs the same solve (LDS reads are broadcasts); lane 0 publishes
                double Hl[36], nb[6], d[6];
                for (int i = 0; i < 36; ++i) Hl[i] = sh.H[i];
                const double lam = sh.lambda;
                for (int i = 0; i < 6; ++i) { Hl[7 * i] += lam; nb[i] = -sh.b[i]; }
                trace_stamp(a.trace, tn, 31);
                if (!solve6(Hl, nb, d)) {
                    if (tid == 0) sh.state = 2;
                } else {
                    trace_stamp(a.trace, tn, 32);
                    double dl[12], x0[12];
                    se3_exp_wave(d, dl, dl + 9, tid);
                    trace_stamp(a.trace, tn, 33);
                    for (int i = 0; i < 12; ++i) x0[i] = sh.x0[i];
                    double den = 0;
                    for (int i = 0; i < 6; ++i) den += d[i] * (lam * d[i] - sh.b[i]);
                    if (tid == 0) {
                        sh.state = 0;
                        for (int i = 0; i < 12; ++i) sh.delta[i] = dl[i];
                        for (int i = 0; i < 3; ++i) {
                            for (int j = 0; j < 3; ++j)
                                sh.xi[3 * i + j] = dl[3 * i] * x0[j] + dl[3 * i + 1] * x0[3 + j] + dl[3 * i + 2] * x0[6 + j];
                            sh.xi[9 + i] = dl[3 * i] * x0[9] + dl[3 * i + 1] * x0[10] + dl[3 * i + 2] * x0[11] + dl[9 + i];
                        }
                        sh.denom = den;
                    }
                }
            }
            __syncthreads();
            trace_stamp(a.trace, tn, 3);   // solve + se3_exp done
            if (sh.state == 2) { failed = 1; break; }
        } else {
            trace_stamp(a.trace, tn, 1);
        }
        // ---------------- the phase: [trial cost with frozen correspondences / Mahalanobis matrices] + linearisation + grid-wide sum
        double both[NRED + 1];
#pragma unroll
        for (int k = 0; k <= NRED; ++k) both[k] = 0;
        if (!opening) {
            double Rx[9], tx[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rx[i] = sh.xi[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) tx[i] = sh.xi[9 + i];
            for (int s = gtid; s < a.n_src; s += gstride) {
                const int c = corr_c[s];
                if (c < 0) continue;
                const float4 p = a.src_pts[a.src_track[s]];
                const float4 bp = a.tgt_pts[c];
                double e[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) e[r] = -(Rx[3 * r] * (double)p.x + Rx[3 * r + 1] * (double)p.y + Rx[3 * r + 2] * (double)p.z + tx[r]);
                e[0] += (double)bp.x; e[1] += (double)bp.y; e[2] += (double)bp.z;
                const double* m = maha_c + 6 * (size_t)s;
                both[NRED] += e[0] * (m[0] * e[0] + m[1] * e[1] + m[2] * e[2]) + e[1] * (m[1] * e[0] + m[3] * e[1] + m[4] * e[2]) +
                              e[2] * (m[2] * e[0] + m[4] * e[1] + m[5] * e[2]);
            }
            trace_stamp(a.trace, tn, 4);   // trial cost done
        }
        linearize_points(a, opening ? sh.x0 : sh.xi, corr_w, sqd_w, maha_w, gtid, gstride, both, s_queue + tid, (a.trace && opening) ? &tn : nullptr);
        trace_stamp(a.trace, tn, opening ? 2 : 5);   // (speculative) linearisation done
        grid_sum<NRED + 1>(both, sh, a.sync, epoch, tid, a.trace, &tn);
        if (sh.abort) { failed = 2; break; }
        if (opening) {
            if (tid < NRED) sh.spec[tid] = sh.red[tid];
            __syncthreads();
            opening = false; trial = 0;
            continue;
        }
        if (tid < 64) {   // first wave: the decision is wave-uniform (every lane reads the same totals), the copies are one element per lane
            const double yi = sh.red[NRED];
            const double rho = (sh.y0 - yi) / sh.denom;
            if (rho < 0) {
                const bool conv = is_converged_wave(sh.delta, sh.delta + 9, a.rot_eps, a.trans_eps, tid);
                if (tid == 0) {
                    if (conv) {
                        sh.state = 1;       // upstream returns true without accepting the step
                    } else {
                        sh.lambda = sh.nu * sh.lambda;
                        sh.nu = 2 * sh.nu;
                        sh.state = 0;
                    }
                }
            } else {
                if (tid < 12) sh.x0[tid] = sh.xi[tid];
                if (tid < NRED) sh.spec[tid] = sh.red[tid];
                if (blockIdx.x == 0 && tid < 36) a.result->H_final[tid] = sh.H[tid];
                if (tid == 0) {
                    const double f = 2 * rho - 1;
                    sh.lambda = sh.lambda * fmax(1.0 / 3.0, 1 - f * f * f);
                    if (leader) a.result->cost = yi;
                    sh.state = 1;
                    sh.accepted = 1;
                }
            }
        }
        __syncthreads();
        if (sh.state != 1) { ++trial; continue; }     // rejected: another trial with a larger damping
        // ---------------- the outer iteration is complete
        ++iterations;
        if (tid < 64) {
            const bool conv = is_converged_wave(sh.delta, sh.delta + 9, a.rot_eps, a.trans_eps, tid);
            if (tid == 0) sh.converged = conv ? 1 : 0;
        }
        __syncthreads();
        if (sh.converged) break;
        const bool have_lin = sh.accepted != 0 && it + 1 < a.max_iter;   // the speculative linearisation of this trial is the next iteration's
        if (have_lin) cur ^= 1;   // the shadow set becomes the set in use; correspondences exported are always those of the set in use
        opening = !have_lin;
        trial = 0;
        ++it;
    }
    trace_stamp(a.trace, tn, 99);
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[511] = (unsigned long long)tn;
    if (cur) {   // leave the exported correspondences / distances in the primary buffers (each thread moves the entries it wrote)
        for (int s = gtid; s < a.n_src; s += gstride) { a.corr[s] = a.corr2[s]; a.sqd[s] = a.sqd2[s]; }
    }
    if (leader) {
        AlignResult* r = a.result;
        for (int i = 0; i < 16; ++i) r->final_pose[i] = (i == 15) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r->final_pose[4 * i + j] = (double)(float)sh.x0[3 * i + j];
            r->final_pose[4 * i + 3] = (double)(float)sh.x0[9 + i];
        }
        r->iterations = iterations; r->lm_trials = lm_trials; r->converged = sh.converged; r->failed = failed;
        // hand the result to the host: plain stores into pinned memory, system-scope fence, then the sequence number
        HostMailbox* mb = a.mailbox;
        const double* src = (const double*)r;
        double* dst = (double*)&mb->result;
        for (unsigned i = 0; i < sizeof(AlignResult) / sizeof(double); ++i) dst[i] = src[i];
        __threadfence_system();
        __hip_atomic_store(&mb->align_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // check-out: a workgroup leaves only after its last barrier, so when the last one leaves nobody reads the barrier
    // state any more and it can be reset for the next launch (no host-side memset between frames)
    __syncthreads();
    if (tid == 0) {
        const unsigned gone = __hip_atomic_fetch_add(&a.sync->exit_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == gridDim.x - 1) {
            __hip_atomic_store(&a.sync->counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.sync->abort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.sync->exit_count, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
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
        if (k < n_src) miss[k] = s;     // (the counter starts at zero: the bound only guards the buffer)
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

// measurement aid (GSICP_TRACKER_EXTRA_LAUNCHES=N): N empty launches in front of the LM kernel price a kernel boundary of the tracker's stream
__global__ void empty_kernel() {}

// Correspondence export straight into pinned host memory (two coalesced streams over PCIe); the last workgroup to finish
// publishes the sequence number the host is polling.
__global__ __launch_bounds__(256) void export_corr_kernel(int n, const int* __restrict__ corr, const float* __restrict__ sqd, int* __restrict__ h_corr,
                                                          float* __restrict__ h_sqd, int* __restrict__ blocks_done, int* __restrict__ miss_counter,
                                                          HostMailbox* mb, unsigned seq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { h_corr[i] = corr[i]; h_sqd[i] = sqd[i]; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(blocks_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(blocks_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the miss list behind this export has been consumed: a second on-demand pass after the same align (gate changed between align and
            // the getter) must start counting from zero again — only the LM kernel zeroed it before (out-of-bounds miss[] writes)
            __hip_atomic_store(miss_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&mb->export_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// set_input_* / set_*_filter ingest: the kernels read the page-locked staging directly (one launch instead of a copy engine
// transfer plus a launch; 132 KB over PCIe is ~5 us).
__global__ __launch_bounds__(256) void ingest_points_kernel(int n, const float4* __restrict__ h_pts, float4* __restrict__ pts, int* __restrict__ track) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { pts[i] = h_pts[i]; track[i] = i; }   // every point trackable until a filter says otherwise
}
// set_input_* immediately followed by set_*_filter — the reference's order [REF mp_Tracker.py:157-158, 191-192] — as ONE launch: the points and,
// for the first n_track rows, the filter's list instead of the identity
__global__ __launch_bounds__(256) void ingest_points_track_kernel(int n, const float4* __restrict__ h_pts, float4* __restrict__ pts, int n_track,
                                                                  const int* __restrict__ h_track, int* __restrict__ track) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { pts[i] = h_pts[i]; track[i] = i < n_track ? h_track[i] : i; }
}
__global__ __launch_bounds__(256) void ingest_track_kernel(int n, const int* __restrict__ h_track, int* __restrict__ track) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) track[i] = h_track[i];
}

// ---- device-side hand-off: clouds and Gaussians that already live in device memory -----------------
__global__ __launch_bounds__(256) void ingest_points_dev_kernel(int n, const float* __restrict__ xyz, float4* __restrict__ pts, int* __restrict__ track) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { pts[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f); track[i] = i; }
}
__global__ __launch_bounds__(256) void copy_f32_kernel(size_t n, const float* __restrict__ src, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// Order-preserving selection of the map's trackable Gaussians [REF scene/gaussian_model.py:207-215]:
// keep i iff opacity[i] > th and trackable_mask[i]; outputs in index order, exactly like torch's boolean indexing.
__device__ inline bool select_keep(int i, int P, const float* __restrict__ opacity, const unsigned char* __restrict__ mask, float th) {
    return i < P && opacity[i] > th && (mask == nullptr || mask[i] != 0);
}
__global__ __launch_bounds__(256) void select_count_kernel(int P, const float* __restrict__ opacity, const unsigned char* __restrict__ mask, float th,
                                                           unsigned* __restrict__ block_count) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(select_keep(i, P, opacity, mask, th));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// single workgroup: exclusive scan of the block counts in place; total -> *total_out and the mailbox
__global__ __launch_bounds__(1024) void select_scan_kernel(int nblocks, unsigned* __restrict__ block_count, unsigned* __restrict__ total_out) {
    __shared__ unsigned s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per) < nblocks ? (lo + per) : nblocks;
    unsigned sum = 0;
    for (int c = lo; c < hi; ++c) sum += block_count[c];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned run = s_part[tid] - sum;
    for (int c = lo; c < hi; ++c) { const unsigned cnt = block_count[c]; block_count[c] = run; run += cnt; }
    if (tid == 1023) *total_out = s_part[1023];
}
__global__ __launch_bounds__(256) void select_scatter_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ rotation,
                                                             const float* __restrict__ scaling, const float* __restrict__ opacity,
                                                             const unsigned char* __restrict__ mask, float th, const unsigned* __restrict__ block_base,
                                                             float4* __restrict__ pts, int* __restrict__ track, float* __restrict__ rotq,
                                                             float* __restrict__ scales) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = select_keep(i, P, opacity, mask, th);
    const unsigned long long b = __ballot(keep);
    if (lane == 0) s_w[wave] = (unsigned)__popcll(b);
    __syncthreads();
    if (!keep) return;
    unsigned pos = block_base[blockIdx.x] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    pts[pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], 0.f);
    track[pos] = (int)pos;
#pragma unroll
    for (int d = 0; d < 4; ++d) rotq[4 * (size_t)pos + d] = rotation[4 * (size_t)i + d];
#pragma unroll
    for (int d = 0; d < 3; ++d) scales[3 * (size_t)pos + d] = scaling[3 * (size_t)i + d];
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

template <class T>
struct PinnedBuf {                // page-locked host staging: async copies without the runtime's own bounce buffer
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        size_t want = n + n / 4 + 64;
        if (hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { cap = 0; return -1; }
        cap = want;
        return 0;
    }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

struct IndexLevel {               // one level of the target index (build_level)
    GridView view{};
    DevBuf<unsigned long long> tkeys;
    DevBuf<unsigned> tords, n_cells, slab_heads;   // n_cells[0] = occupied coarse cells (grid_heads_scan_kernel)
    DevBuf<uint2> cells;
    DevBuf<float4> sorted;
    unsigned n_cells_host = 0;
    bool cells_known = false;     // n_cells_host was read back (else it is the upper bound n)
};

struct Cloud {
    int n = 0, n_track = 0;
    DevBuf<float4> pts;
    DevBuf<int> track;
    PinnedBuf<float4> h_pts;      // staging for set_input_*; reused only after `staged` has completed
    PinnedBuf<int> h_track;
    hipEvent_t staged[2] = {nullptr, nullptr};   // [0] h_pts copy, [1] h_track copy
    bool staged_pending[2] = {false, false};
    bool pts_pending = false;     // h_pts holds a cloud whose ingest launch is deferred to the filter call (one fused launch) or to the first consumer
    ~Cloud() { for (hipEvent_t e : staged) if (e) (void)hipEventDestroy(e); }
    DevBuf<double> cov;
    DevBuf<float> rotq, scales;
    bool cov_valid = false, qs_valid = false;
};

}  // namespace

struct gsicp_gicp {
    hipStream_t stream = nullptr;
    int k = 20, max_iter = 64, lm_max_iter = 10, reg = 3;
    int scale_mode = 0;   // exported scales / fromqs input: 0 = std-devs (sqrt eigenvalues; fromqs squares them), 1 = variances (eigenvalues; used as they are)
    double max_corr = (double)FLT_MAX, max_knn = (double)FLT_MAX, rot_eps = 2e-3, trans_eps = 5e-4, lm_init = 1e-9;
    Cloud src, tgt;
    // target search structure
    bool grid_valid = false;
    IndexLevel lv[2];                      // [0] complete within the gate; [1] dense maps only: complete within sqrt(fine_r2), tried first
    float fine_r2 = 0.f;
    DevBuf<unsigned long long> gkeys, gskeys, packed;
    DevBuf<unsigned> gvals, gsvals;
    DevBuf<char> sort_temp;
    // per-source-point outputs
    DevBuf<int> corr, corr2, miss, counters, nbr_idx, knn_cell_of;
    DevBuf<KnnGrid> knn_params;
    DevBuf<unsigned> knn_count, knn_start, knn_fill;
    DevBuf<float4> knn_sorted;
    // the same kind of grid over the trackable TARGET points (built with the hashed grid, keyframe rate): exact 1-NN export
    DevBuf<KnnGrid> tg_params;
    DevBuf<unsigned> tg_count, tg_start, tg_fill;
    DevBuf<float4> tg_sorted;
    DevBuf<int> tg_cell_of;
    bool tg_valid = false;
    DevBuf<float> nbr_d2;
    DevBuf<unsigned long long> trace;
    DevBuf<float> sqd, sqd2;
    DevBuf<double> maha, maha2;
    DevBuf<AlignResult> result;
    DevBuf<AlignSync> sync;
    HostMailbox* mailbox = nullptr;        // pinned
    unsigned seq = 0;
    bool stats_pending = false;            // device_us of the last align not read from the events yet
    int max_resident_wg = 0;               // occupancy bound of the persistent align kernel on this device (grid barrier needs co-residency)
    bool inject_abort = false;             // test hook: make the next align's first barrier abort (gsicp_gicp_debug_abort_next_align)
    int barrier_retries = 0;               // aligns that lost their grid barrier and were re-run as one workgroup
    PinnedBuf<int> h_corr;
    PinnedBuf<float> h_sqd;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr, ev_xs = nullptr, ev_xs2 = nullptr;
    DevBuf<unsigned> sel_blocks;
    AlignResult host_result{};
    bool aligned = false, dist_exact = false;
    // get_source_correspondence's kernels enqueued by align itself, right behind the LM kernel (see enqueue_correspondence_export): switched on
    // by the first get_source_correspondence call of this object — the reference asks after every align [REF mp_Tracker.py:231] — so a caller
    // that never asks never pays for them
    bool spec_corr = false, spec_valid = false;
    int spec_valid_unfetched = 0;          // aligns since the last get_source_correspondence (8 in a row switch the speculation off again)
    unsigned spec_seq = 0;
    int spec_m = 0;
    std::vector<float> h_stage;
    double stats[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

// Wait for the tracker stream by polling an event: the runtime's hipStreamSynchronize parks the thread and its wake-up costs
// 50-150 us, several times the kernels being waited for.  Falls back to the blocking wait after ~2 ms of polling.
int drain(gsicp_gicp* g) {
    if (!g->ev_done) GC(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
    GC(hipEventRecord(g->ev_done, g->stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(g->ev_done);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { g_last_error = std::string("hipEventQuery failed: ") + hipGetErrorString(q); return -1; }
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    GC(hipStreamSynchronize(g->stream));
    return 0;
}

// Poll a sequence number a kernel publishes in the pinned mailbox; after ~2 ms fall back to draining the stream (and fail if
// the number still is not there — the launch itself must have failed).
int wait_mailbox(gsicp_gicp* g, const volatile unsigned* slot, unsigned seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (int rc = drain(g)) return rc;
    if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == seq) return 0;
    g_last_error = "tracker kernel finished without publishing its result";
    return -1;
}

int wait_staging(Cloud& c, int which) {
    if (c.staged_pending[which]) { GC(hipEventSynchronize(c.staged[which])); c.staged_pending[which] = false; }
    return 0;
}
int mark_staging(gsicp_gicp* g, Cloud& c, int which) {
    if (!c.staged[which]) GC(hipEventCreateWithFlags(&c.staged[which], hipEventDisableTiming));
    GC(hipEventRecord(c.staged[which], g->stream));
    c.staged_pending[which] = true;
    return 0;
}

// The caller's array is converted into page-locked staging and copied asynchronously; the call returns without waiting for
// the device (the caller's buffer is already consumed), and the staging is reused only once its copy has completed.
int upload_points(gsicp_gicp* g, Cloud& c, const void* pts, int n, int is_f64) {
    if (n < 0 || (n > 0 && !pts)) { g_last_error = "bad point array"; return -2; }
    c.n = n; c.n_track = n; c.cov_valid = false; c.qs_valid = false;
    if (c.pts.ensure((size_t)n) || c.track.ensure((size_t)n) || c.h_pts.ensure((size_t)n)) { g_last_error = "hipMalloc failed"; return -1; }
    if (n > 0) {
        if (int rc = wait_staging(c, 0)) return rc;
        float4* h = c.h_pts.p;
        if (is_f64) { const double* p = (const double*)pts; for (int i = 0; i < n; ++i, p += 3) h[i] = make_float4((float)p[0], (float)p[1], (float)p[2], 0.f); }
        else { const float* p = (const float*)pts; for (int i = 0; i < n; ++i, p += 3) h[i] = make_float4(p[0], p[1], p[2], 0.f); }
        c.pts_pending = true;      // launched by upload_filter (fused with the filter's list) or by flush_points
    } else c.pts_pending = false;
    return 0;
}

// Every consumer of a cloud's device arrays calls this first: a cloud that was uploaded without a filter call behind it is ingested now.
int flush_points(gsicp_gicp* g, Cloud& c) {
    if (!c.pts_pending) return 0;
    c.pts_pending = false;
    hipLaunchKernelGGL(ingest_points_kernel, dim3((c.n + 255) / 256), dim3(256), 0, g->stream, c.n, (const float4*)c.h_pts.p, c.pts.p, c.track.p);
    return mark_staging(g, c, 0);
}

int upload_filter(gsicp_gicp* g, Cloud& c, int n_track, const int32_t* f, int n) {
    if (n != c.n) { g_last_error = "filter length does not match the point cloud"; return -2; }
    const size_t nt = (size_t)(n_track > 0 ? n_track : 0);
    if (c.h_track.ensure(nt ? nt : 1)) { g_last_error = "hipHostMalloc failed"; return -1; }
    if (int rc = wait_staging(c, 1)) return rc;
    int* tr = c.h_track.p;
    for (size_t i = 0; i < nt; ++i) tr[i] = -1;
    for (int i = 0; i < n; ++i)
        if (f[i] > 0 && f[i] <= n_track) tr[f[i] - 1] = i;
    size_t w = 0;
    for (size_t i = 0; i < nt; ++i)
        if (tr[i] >= 0) tr[w++] = tr[i];
    c.n_track = (int)w;
    if (c.track.ensure(w ? w : 1)) { g_last_error = "hipMalloc failed"; return -1; }
    if (c.pts_pending && n > 0) {
        c.pts_pending = false;
        hipLaunchKernelGGL(ingest_points_track_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, (const float4*)c.h_pts.p, c.pts.p, (int)w,
                           (const int*)tr, c.track.p);
        if (int rc = mark_staging(g, c, 0)) return rc;
        if (int rc = mark_staging(g, c, 1)) return rc;
    } else if (w) {
        hipLaunchKernelGGL(ingest_track_kernel, dim3(((int)w + 255) / 256), dim3(256), 0, g->stream, (int)w, (const int*)tr, c.track.p);
        if (int rc = mark_staging(g, c, 1)) return rc;
    }
    return 0;
}

int calc_cov(gsicp_gicp* g, Cloud& c) {
    const int n = c.n;
    if (int rc = flush_points(g, c)) return rc;
    if (c.cov.ensure((size_t)6 * (n ? n : 1)) || c.rotq.ensure((size_t)4 * (n ? n : 1)) || c.scales.ensure((size_t)3 * (n ? n : 1))) {
        g_last_error = "hipMalloc failed"; return -1;
    }
    if (n > 0) {
        if (g->k > 64) { g_last_error = "correspondence randomness (k) > 64 is not supported"; return -2; }
        const float maxd2 = g->max_knn >= (double)FLT_MAX ? FLT_MAX : (float)(g->max_knn * g->max_knn);
        gsicp::ProfileScope ps(gsicp::ST_GICP_COV, g->stream);
        if (g->nbr_idx.ensure((size_t)n * 64) || g->nbr_d2.ensure((size_t)n * 64) || g->knn_params.ensure(KNN_PARAM_SLOTS) ||
            g->knn_cell_of.ensure((size_t)n) || g->knn_count.ensure(KNN_CELL_SLOTS) || g->knn_start.ensure(KNN_CELL_SLOTS) ||
            g->knn_fill.ensure(KNN_CELL_SLOTS) || g->knn_sorted.ensure((size_t)n)) { g_last_error = "hipMalloc failed"; return -1; }
        static const bool knn_stats_on = std::getenv("GSICP_KNN_STATS") != nullptr;
        static const float h_area = [] { const char* e = std::getenv("GSICP_KNN_H"); const float v = e ? (float)std::atof(e) : 0.f; return v > 0.f ? v : KNN_H_AREA; }();
        if (n <= KNN_FUSED_MAX_N) {
            hipLaunchKernelGGL(knn_build_fused_kernel, dim3(1), dim3(1024), 0, g->stream, n, c.pts.p, h_area, h_area * (KNN_H_VOL / KNN_H_AREA),
                               g->knn_params.p, g->knn_cell_of.p, g->knn_count.p, g->knn_start.p, g->knn_fill.p, g->knn_sorted.p);
        } else {
            enqueue_knn_grid_build(g->stream, n, c.pts.p, h_area, h_area * (KNN_H_VOL / KNN_H_AREA), g->knn_params.p, g->knn_cell_of.p, g->knn_count.p,
                                   g->knn_start.p, g->knn_fill.p, g->knn_sorted.p);
        }
        hipLaunchKernelGGL(knn_grid_kernel, dim3((n + 3) / 4), dim3(256), 0, g->stream, n, g->k, c.pts.p, g->knn_params.p, g->knn_start.p,
                           g->knn_sorted.p, g->nbr_idx.p, g->nbr_d2.p,
                           knn_stats_on ? g->knn_params.p->ring_hist : (unsigned*)nullptr);   // same-address atomics: diagnostics only
        hipLaunchKernelGGL(cov_eig_kernel, dim3((n + 63) / 64), dim3(64), 0, g->stream, n, g->k, c.pts.p, g->nbr_idx.p, g->nbr_d2.p, maxd2,
                           g->reg, g->scale_mode, c.cov.p, c.rotq.p, c.scales.p);
        GC(hipGetLastError());
    }
    c.cov_valid = true; c.qs_valid = true;
    return 0;
}

// One level of the target index: points sorted by (coarse cell, octant), hashed coarse cells, eight {begin, count} pairs per cell, complete
// within `radius` of any query.  Targets of >= 32 768 points get their table and cell records sized by the number of OCCUPIED cells (one
// 4-byte read-back: this runs at keyframe rate, and the hand-off that precedes it synchronises anyway): a map has 6-60 points per cell, so
// the table shrinks from 48 MB to ~3 MB at 1e6 points and the 8 bucket probes of a query come from the L2 instead of HBM.
int build_level(gsicp_gicp* g, int lvl, double radius) {
    Cloud& t = g->tgt;
    const int n = t.n_track;
    const int slab = n < GRID_SMALL_N ? 256 : 2048, n_slabs = (n + slab - 1) / slab;
    IndexLevel& L = g->lv[lvl];
    GridView& G = L.view;
    std::memset(&G, 0, sizeof(G));
    if (L.sorted.ensure((size_t)n)) { g_last_error = "hipMalloc failed"; return -1; }
    // fine cell edge = radius x 1.001: every target within the radius of a query lies in the 27 fine cells around the query's (the 0.1 %
    // covers the float rounding of the cell assignment for coordinates up to ~80 m at a 2 cm radius); hashed coarse cell = 2x2x2 fine cells
    const float hf = (float)(radius * 1.001);
    G.use_grid = 1; G.hf = hf; G.inv_hf = 1.0f / hf; G.sorted = L.sorted.p; G.n_sorted = n;
    size_t temp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, (size_t)n, 0, 63, g->stream);
    if (g->gkeys.ensure(n) || g->gskeys.ensure(n) || g->gvals.ensure(n) || g->gsvals.ensure(n) || L.n_cells.ensure(2) || L.slab_heads.ensure((size_t)n_slabs) ||
        g->sort_temp.ensure(temp_bytes ? temp_bytes : 1)) { g_last_error = "hipMalloc failed"; return -1; }
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(grid_keys_kernel, grid, block, 0, g->stream, n, t.track.p, t.pts.p, G.inv_hf, g->gkeys.p, g->gvals.p);
    GC(rocprim::radix_sort_pairs(g->sort_temp.p, temp_bytes, g->gkeys.p, g->gskeys.p, g->gvals.p, g->gsvals.p, (size_t)n, 0, 63, g->stream));
    if (slab == 256) hipLaunchKernelGGL(grid_heads_kernel<1>, dim3(n_slabs), block, 0, g->stream, n, g->gskeys.p, L.slab_heads.p);
    else hipLaunchKernelGGL(grid_heads_kernel<8>, dim3(n_slabs), block, 0, g->stream, n, g->gskeys.p, L.slab_heads.p);
    hipLaunchKernelGGL(grid_heads_scan_kernel, dim3(1), dim3(1024), 0, g->stream, n_slabs, L.slab_heads.p, L.n_cells.p);
    size_t n_cells = (size_t)n;       // upper bound
    L.cells_known = false;
    if (n >= (1 << 15)) {
        GC(hipMemcpyAsync(&g->mailbox->scratch_u32, L.n_cells.p, sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
        if (int rc_ = drain(g)) return rc_;
        n_cells = g->mailbox->scratch_u32;
        if (n_cells < 1 || n_cells > (size_t)n) { g_last_error = "target index: bad cell count"; return -1; }
        L.cells_known = true;
    }
    L.n_cells_host = (unsigned)n_cells;
    // load factor <= 1/8 (in cells when counted, in points otherwise — a depth frame has ~1.5 points per cell): most of the 8 cells a
    // query probes do not exist, and an unsuccessful linear probe costs ~(1 + 1/(1-a)^2)/2 dependent loads ON AVERAGE but the slowest lane
    // of a wave sets the pace — at a = 1/2 the 8-cell search took 7 us of every phase
    size_t cap = 64;
    while (cap < (size_t)8 * n_cells) cap <<= 1;
    G.mask = (unsigned)(cap - 1);
    if (L.tkeys.ensure(cap) || L.tords.ensure(cap) || L.cells.ensure((size_t)8 * n_cells)) { g_last_error = "hipMalloc failed"; return -1; }
    GC(hipMemsetAsync(L.tkeys.p, 0xFF, cap * 8, g->stream));
    if (slab == 256) hipLaunchKernelGGL(grid_cells_kernel<1>, dim3(n_slabs), block, 0, g->stream, n, g->gskeys.p, g->gsvals.p, t.pts.p, L.sorted.p, G.mask, L.tkeys.p, L.tords.p,
                       L.cells.p, L.slab_heads.p);
    else hipLaunchKernelGGL(grid_cells_kernel<8>, dim3(n_slabs), block, 0, g->stream, n, g->gskeys.p, g->gsvals.p, t.pts.p, L.sorted.p, G.mask, L.tkeys.p, L.tords.p,
                       L.cells.p, L.slab_heads.p);
    hipLaunchKernelGGL(grid_octants_kernel, grid, block, 0, g->stream, n, g->gskeys.p, G.mask, L.tkeys.p, L.tords.p, L.cells.p);
    GC(hipGetLastError());
    G.keys = L.tkeys.p; G.ords = L.tords.p; G.cells = L.cells.p;
    return 0;
}

int build_grid(gsicp_gicp* g) {
    Cloud& t = g->tgt;
    if (int rc = flush_points(g, t)) return rc;
    const int n = t.n_track;
    GridView& G = g->lv[0].view;
    std::memset(&G, 0, sizeof(G));
    std::memset(&g->lv[1].view, 0, sizeof(GridView));
    g->fine_r2 = 0.f;
    g->tg_valid = false;
    if (g->lv[0].sorted.ensure((size_t)(n ? n : 1))) { g_last_error = "hipMalloc failed"; return -1; }
    G.sorted = g->lv[0].sorted.p; G.n_sorted = n;
    gsicp::ProfileScope ps(gsicp::ST_GICP_GRID, g->stream);
    const bool gated = g->max_corr < 1e6 && g->max_corr > 0;
    if (!gated || n == 0) {
        G.use_grid = 0;
        if (n > 0) hipLaunchKernelGGL(gather_track_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, t.track.p, t.pts.p, g->lv[0].sorted.p);
        GC(hipGetLastError());
        g->grid_valid = true;
        return 0;
    }
    if (n >= (1 << 26)) { g_last_error = "target index: more than 2^26 trackable targets (a work-list entry packs a cell's count in 26 bits)"; return -2; }
    if (int rc = build_level(g, 0, g->max_corr)) return rc;
    // EXTREMELY dense maps (>= 64 points per occupied gate-sized coarse cell) get a second level of the same structure at a smaller radius
    // r1 = f x gate, tried first (linearize_points): a neighbour found inside r1 is the nearest neighbour.  With m points per coarse cell the
    // targets lie on surfaces at ~m / (2 gate)^2 per unit area; f = sqrt(3.8 / m) puts ~3 expected points inside r1 and ~1 into a fine cell
    // of the level.  Measured (MI355X, 8 280-point frame 7 mm from its pose, align kernel us, one level | two): m = 17.6 (3e5 Gaussians on
    // one room corner's surfaces) 109 | 138, m = 57 (1e6) 203 | 190, m = 171 (3e6) 425 | 303 — a query that starts several mm off the
    // surface falls through to the gate-sized level in the first linearisation, so the second level only pays where cells are very full; the
    // build costs +25-35 %.  A whole-room Replica map of 1-2 M Gaussians has m ~ 16-32: one level.
    // A/B switches, read at every build (keyframe rate): GSICP_INDEX_LEVELS=1 keeps one level; GSICP_INDEX_L2_MIN_FILL moves the threshold (tests)
    const char* lv_env = std::getenv("GSICP_INDEX_LEVELS");
    const char* fill_env = std::getenv("GSICP_INDEX_L2_MIN_FILL");
    const int two_level = lv_env ? std::atoi(lv_env) : 2;
    const double min_fill = fill_env ? std::atof(fill_env) : 64.0;
    if (g->lv[0].cells_known && two_level >= 2) {
        const double m = (double)n / (double)g->lv[0].n_cells_host;
        double f = std::sqrt(3.8 / m);
        if (m >= min_fill) {
            if (f < 0.15) f = 0.15;
            const double r1 = g->max_corr * f;
            if (int rc = build_level(g, 1, r1)) return rc;
            g->fine_r2 = (float)r1 * (float)r1;
        }
    }
    // coarse dense grid over the same points for the exact-distance export (nn1_grid_kernel)
    g->tg_valid = false;
    if (!(g->tg_params.ensure(KNN_PARAM_SLOTS) || g->tg_cell_of.ensure((size_t)n) || g->tg_count.ensure(KNN_CELL_SLOTS) ||
          g->tg_start.ensure(KNN_CELL_SLOTS) || g->tg_fill.ensure(KNN_CELL_SLOTS) || g->tg_sorted.ensure((size_t)n))) {
        if (n <= KNN_FUSED_MAX_N)      // a frame-sized target (the first frames of a run): the single-workgroup build, one launch instead of six
            hipLaunchKernelGGL(knn_build_fused_kernel, dim3(1), dim3(1024), 0, g->stream, n, (const float4*)g->lv[0].sorted.p, KNN_H_AREA, KNN_H_VOL,
                               g->tg_params.p, g->tg_cell_of.p, g->tg_count.p, g->tg_start.p, g->tg_fill.p, g->tg_sorted.p);
        else
            enqueue_knn_grid_build(g->stream, n, (const float4*)g->lv[0].sorted.p, KNN_H_AREA, KNN_H_VOL, g->tg_params.p, g->tg_cell_of.p, g->tg_count.p,
                                   g->tg_start.p, g->tg_fill.p, g->tg_sorted.p);
        GC(hipGetLastError());
        g->tg_valid = true;
    }
    g->grid_valid = true;
    return 0;
}

int fetch_floats(gsicp_gicp* g, const float* dev, int n_pts, int width, float* out, int cap_pts) {
    const int n = n_pts < cap_pts ? n_pts : cap_pts;
    if (n > 0) {
        GC(hipMemcpyAsync(out, dev, sizeof(float) * (size_t)n * width, hipMemcpyDeviceToHost, g->stream));
        if (int rc_ = drain(g)) return rc_;
    }
    return n;
}

}  // namespace

extern "C" {

void* gsicp_stream_create_cu_mask(int first_cu, int n_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { g_last_error = "gsicp_stream_create_cu_mask: no HIP device"; return nullptr; }
    const int total = prop.multiProcessorCount;
    if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total) { g_last_error = "gsicp_stream_create_cu_mask: mask bits outside the device's compute units"; return nullptr; }
    std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
    for (int b = first_cu; b < first_cu + n_cus; ++b) mask[(size_t)b >> 5] |= 1u << (b & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        g_last_error = "hipExtStreamCreateWithCUMask failed";
        return nullptr;
    }
    return (void*)s;
}
int gsicp_stream_destroy(void* stream) {
    if (!stream) return 0;
    (void)hipStreamSynchronize((hipStream_t)stream);
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : -1;
}

gsicp_gicp* gsicp_gicp_create(void) {
    gsicp_gicp* g = new gsicp_gicp();
    // The tracker is the latency-critical half (one frame = a short dependent chain of small kernels); the mapper it shares the GPU
    // with is throughput work.  Highest stream priority lets the tracker's waves go ahead of queued mapper workgroups.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char* pe = std::getenv("GSICP_TRACKER_PRIORITY");   // "0" = default priority (A/B switch for measurements)
    if (pe && pe[0] == '0') prio_hi = prio_lo = 0;
    // GSICP_TRACKER_CU_MASK="first:count" (round 6 experiment, default off): the tracker's stream on `count` dedicated compute units starting at mask bit
    // `first` (hipExtStreamCreateWithCUMask; on a multi-XCD part consecutive mask bits go round the XCDs, so a prefix is spread evenly over the eight) —
    // its 33 workgroups then never queue behind mapper waves on those CUs when the mapper's stream carries the complement (gsicp_stream_create_cu_mask).
    // A masked stream has no priority argument: the two are alternatives.
    const char* cm = std::getenv("GSICP_TRACKER_CU_MASK");
    int cu_first = 0, cu_count = 0;
    if (cm && std::sscanf(cm, "%d:%d", &cu_first, &cu_count) == 2 && cu_count > 0) {
        g->stream = (hipStream_t)gsicp_stream_create_cu_mask(cu_first, cu_count);
        if (!g->stream) { delete g; return nullptr; }
    } else if (hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
        g_last_error = "hipStreamCreate failed (no HIP device?)";
        delete g;
        return nullptr;
    }
    if (g->result.ensure(1) || g->counters.ensure(4) || g->sync.ensure(1) ||
        hipHostMalloc((void**)&g->mailbox, sizeof(HostMailbox), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess ||
        hipMemset(g->sync.p, 0, 128) != hipSuccess || hipMemset(g->counters.p, 0, sizeof(int) * 4) != hipSuccess) {
        g_last_error = "device / pinned allocation failed"; gsicp_gicp_destroy(g); return nullptr;
    }
    std::memset(g->mailbox, 0, sizeof(HostMailbox));
    {   // the align kernel's grid barrier needs every workgroup resident at once: bound the grid by what this device can hold
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gicp_align_kernel, AL_T, 0) == hipSuccess && per_cu > 0 &&
            hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            g->max_resident_wg = per_cu * prop.multiProcessorCount;
        else
            g->max_resident_wg = 1;   // unknown occupancy: a single workgroup needs no grid barrier
    }
    return g;
}
void gsicp_gicp_destroy(gsicp_gicp* g) {
    if (!g) return;
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    if (g->mailbox) (void)hipHostFree(g->mailbox);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->ev_done) (void)hipEventDestroy(g->ev_done);
    if (g->ev_xs) (void)hipEventDestroy(g->ev_xs);
    if (g->ev_xs2) (void)hipEventDestroy(g->ev_xs2);
    delete g;
}
int gsicp_gicp_set_max_correspondence_distance(gsicp_gicp* g, double d) { g->max_corr = d; g->grid_valid = false; g->spec_valid = false; g->dist_exact = false; return 0; }
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
int gsicp_gicp_set_scale_semantics(gsicp_gicp* g, int mode) {
    if (mode < 0 || mode > 1) { g_last_error = "scale semantics: 0 = std-dev, 1 = variance"; return -2; }
    g->scale_mode = mode; g->src.cov_valid = false; g->tgt.cov_valid = false; return 0;
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
    if (rc == 0) if (int rc_ = drain(g)) return rc_;
    return rc;
}
int gsicp_gicp_calculate_source_covariance(gsicp_gicp* g) {
    const int rc = calc_cov(g, g->src);
    if (rc == 0) if (int rc_ = drain(g)) return rc_;
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
        hipLaunchKernelGGL(cov_fromqs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, g->stream, t.n, g->reg, g->scale_mode, t.rotq.p, t.scales.p, t.cov.p);
        GC(hipGetLastError());
        if (int rc_ = drain(g)) return rc_;
    }
    t.cov_valid = true; t.qs_valid = true;
    return 0;
}

// ---- device-pointer overloads (additive; the numpy API above is unchanged) --------------------------------------------------
namespace {
// order the tracker stream after whatever produced the caller's device buffers
int wait_for_producer(gsicp_gicp* g, void* producer_stream) {
    if (!g->ev_xs) GC(hipEventCreateWithFlags(&g->ev_xs, hipEventDisableTiming));
    GC(hipEventRecord(g->ev_xs, (hipStream_t)producer_stream));
    GC(hipStreamWaitEvent(g->stream, g->ev_xs, 0));
    return 0;
}
int upload_points_device(gsicp_gicp* g, Cloud& c, const float* xyz, int n, void* producer_stream, int wait) {
    c.pts_pending = false;     // a host upload that was never consumed is superseded
    if (n < 0 || (n > 0 && !xyz)) { g_last_error = "bad device point array"; return -2; }
    c.n = n; c.n_track = n; c.cov_valid = false; c.qs_valid = false;
    if (c.pts.ensure((size_t)n) || c.track.ensure((size_t)n)) { g_last_error = "hipMalloc failed"; return -1; }
    if (n > 0) {
        if (int rc = wait_for_producer(g, producer_stream)) return rc;
        hipLaunchKernelGGL(ingest_points_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, xyz, c.pts.p, c.track.p);
        GC(hipGetLastError());
        if (wait) { if (int rc_ = drain(g)) return rc_; }
    }
    return 0;
}
}  // namespace

void* gsicp_gicp_stream(gsicp_gicp* g) { return (void*)g->stream; }

int gsicp_gicp_set_input_target_device(gsicp_gicp* g, const float* xyz, int n, void* producer_stream, int wait) {
    g->grid_valid = false; g->aligned = false;
    return upload_points_device(g, g->tgt, xyz, n, producer_stream, wait);
}
int gsicp_gicp_set_input_source_device(gsicp_gicp* g, const float* xyz, int n, void* producer_stream, int wait) {
    g->aligned = false;
    return upload_points_device(g, g->src, xyz, n, producer_stream, wait);
}
int gsicp_gicp_set_target_covariances_fromqs_device(gsicp_gicp* g, const float* rots, int n_rots, const float* scales, int n_scales,
                                                    void* producer_stream, int wait) {
    Cloud& t = g->tgt;
    if (n_rots != 4 * t.n || n_scales != 3 * t.n) { g_last_error = "rotations / scales do not match the target cloud size"; return -2; }
    const size_t n = (size_t)(t.n ? t.n : 1);
    if (t.cov.ensure(6 * n) || t.rotq.ensure(4 * n) || t.scales.ensure(3 * n)) { g_last_error = "hipMalloc failed"; return -1; }
    if (t.n > 0) {
        if (int rc = wait_for_producer(g, producer_stream)) return rc;
        hipLaunchKernelGGL(copy_f32_kernel, dim3((4 * t.n + 255) / 256), dim3(256), 0, g->stream, (size_t)4 * t.n, rots, t.rotq.p);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((3 * t.n + 255) / 256), dim3(256), 0, g->stream, (size_t)3 * t.n, scales, t.scales.p);
        hipLaunchKernelGGL(cov_fromqs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, g->stream, t.n, g->reg, g->scale_mode, t.rotq.p, t.scales.p, t.cov.p);
        GC(hipGetLastError());
        if (wait) { if (int rc_ = drain(g)) return rc_; }
    }
    t.cov_valid = true; t.qs_valid = true;
    return 0;
}
int gsicp_gicp_set_target_from_gaussians_device(gsicp_gicp* g, int P, const float* xyz, const float* rotation, const float* scaling,
                                                const float* opacity, const unsigned char* trackable_mask, float opacity_th,
                                                void* producer_stream) {
    if (P < 0 || (P > 0 && (!xyz || !rotation || !scaling || !opacity))) { g_last_error = "set_target_from_gaussians: bad arguments"; return -2; }
    Cloud& t = g->tgt;
    g->grid_valid = false; g->aligned = false;
    t.n = 0; t.n_track = 0; t.cov_valid = false; t.qs_valid = false; t.pts_pending = false;
    if (P == 0) return 0;
    const int nblocks = (P + 255) / 256;
    const size_t cap = (size_t)P;
    if (t.pts.ensure(cap) || t.track.ensure(cap) || t.cov.ensure(6 * cap) || t.rotq.ensure(4 * cap) || t.scales.ensure(3 * cap) ||
        g->sel_blocks.ensure((size_t)nblocks + 1)) { g_last_error = "hipMalloc failed"; return -1; }
    if (int rc = wait_for_producer(g, producer_stream)) return rc;
    unsigned* total_dev = g->sel_blocks.p + nblocks;
    hipLaunchKernelGGL(select_count_kernel, dim3(nblocks), dim3(256), 0, g->stream, P, opacity, trackable_mask, opacity_th, g->sel_blocks.p);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, g->stream, nblocks, g->sel_blocks.p, total_dev);
    hipLaunchKernelGGL(select_scatter_kernel, dim3(nblocks), dim3(256), 0, g->stream, P, xyz, rotation, scaling, opacity, trackable_mask, opacity_th,
                       g->sel_blocks.p, t.pts.p, t.track.p, t.rotq.p, t.scales.p);
    GC(hipGetLastError());
    unsigned total = 0;
    GC(hipMemcpyAsync(&g->mailbox->scratch_u32, total_dev, sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
    if (int rc_ = drain(g)) return rc_;
    total = g->mailbox->scratch_u32;
    t.n = (int)total; t.n_track = (int)total;
    if (total > 0) {
        hipLaunchKernelGGL(cov_fromqs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, g->stream, t.n, g->reg, g->scale_mode, t.rotq.p, t.scales.p, t.cov.p);
        GC(hipGetLastError());
    }
    t.cov_valid = true; t.qs_valid = true;
    return (int)total;
}
int gsicp_gicp_set_source_track_device(gsicp_gicp* g, const int* trackable_idx, int n_track, void* producer_stream, int wait) {
    Cloud& c = g->src;
    if (n_track < 0 || n_track > c.n || (n_track > 0 && !trackable_idx)) { g_last_error = "set_source_track: bad list"; return -2; }
    if (int rc = flush_points(g, c)) return rc;     // a host-uploaded cloud writes its identity list first
    c.n_track = n_track;
    if (c.track.ensure((size_t)(n_track ? n_track : 1))) { g_last_error = "hipMalloc failed"; return -1; }
    if (n_track > 0) {
        if (int rc = wait_for_producer(g, producer_stream)) return rc;
        hipLaunchKernelGGL(ingest_track_kernel, dim3((n_track + 255) / 256), dim3(256), 0, g->stream, n_track, trackable_idx, c.track.p);
        GC(hipGetLastError());
        if (wait) { if (int rc_ = drain(g)) return rc_; }
    }
    g->aligned = false;
    return 0;
}
static int fetch_floats_device(gsicp_gicp* g, const float* dev, int n_pts, int width, float* out_dev, int cap_pts, void* consumer_stream) {
    const int n = n_pts < cap_pts ? n_pts : cap_pts;
    if (n > 0) {
        if (!out_dev) { g_last_error = "null device output"; return -2; }
        const size_t cnt = (size_t)n * width;
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, g->stream, cnt, dev, out_dev);
        GC(hipGetLastError());
        if (!g->ev_xs2) GC(hipEventCreateWithFlags(&g->ev_xs2, hipEventDisableTiming));
        GC(hipEventRecord(g->ev_xs2, g->stream));
        GC(hipStreamWaitEvent((hipStream_t)consumer_stream, g->ev_xs2, 0));   // the consumer's stream sees the data; no host wait
    }
    return n;
}
int gsicp_gicp_get_source_rotationsq_device(gsicp_gicp* g, float* out_dev, int cap, void* consumer_stream) {
    if (!g->src.qs_valid) { if (int rc = calc_cov(g, g->src)) return rc; }
    return fetch_floats_device(g, g->src.rotq.p, g->src.n, 4, out_dev, cap, consumer_stream);
}
int gsicp_gicp_get_source_scales_device(gsicp_gicp* g, float* out_dev, int cap, void* consumer_stream) {
    if (!g->src.qs_valid) { if (int rc = calc_cov(g, g->src)) return rc; }
    return fetch_floats_device(g, g->src.scales.p, g->src.n, 3, out_dev, cap, consumer_stream);
}

// The kernels behind get_source_correspondence [REF mp_Tracker.py:231]: exact distances for the points the gated search left without a match
// (miss list -> exact nearest neighbour), then the export of (index, squared distance) into page-locked staging, published through the mailbox.
// Everything they read is on the device (the final correspondences, the pose of the last linearisation), so they can be enqueued right behind
// the LM kernel without waiting for it: `gsicp_gicp_align` does that once a caller has shown that it asks for correspondences, which takes the
// host round trip (align returns -> Python -> three launches, each waiting for CUs next to the mapper) out of the frame.
static int enqueue_correspondence_export(gsicp_gicp* g, int m, unsigned* seq_out) {
    Cloud &s = g->src, &t = g->tgt;
    const int n = s.n_track;
    if (!g->dist_exact && g->lv[0].view.use_grid && n > 0 && t.n_track > 0) {
        const float gate = (float)g->max_corr * (float)g->max_corr;
        gsicp::ProfileScope ps(gsicp::ST_GICP_MISS, g->stream);
        hipLaunchKernelGGL(miss_list_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, n, g->sqd.p, g->corr.p, gate, g->miss.p,
                           g->counters.p);
        if (g->tg_valid)
            hipLaunchKernelGGL(nn1_grid_kernel, dim3((n + 3) / 4), dim3(256), 0, g->stream, g->miss.p, g->counters.p, s.track.p, s.pts.p,
                               g->result.p->lin_pose, g->tg_params.p, g->tg_start.p, g->tg_sorted.p, t.n_track, g->sqd.p);
        else
            hipLaunchKernelGGL(brute_nn_kernel, dim3((n + 3) / 4), dim3(256), 0, g->stream, g->miss.p, g->counters.p, s.track.p, s.pts.p,
                               g->result.p->lin_pose, g->lv[0].sorted.p, t.n_track, g->sqd.p);
        GC(hipGetLastError());
        g->dist_exact = true;
    }
    if ((size_t)m > g->h_corr.cap || (size_t)m > g->h_sqd.cap) {     // growing frees the old staging: an export an earlier align enqueued (and nobody
        if (int rc = drain(g)) return rc;                            // fetched) may still be writing it
    }
    if (g->h_corr.ensure((size_t)m) || g->h_sqd.ensure((size_t)m)) { g_last_error = "hipHostMalloc failed"; return -1; }
    const unsigned seq = ++g->seq;
    hipLaunchKernelGGL(export_corr_kernel, dim3((m + 255) / 256), dim3(256), 0, g->stream, m, g->corr.p, g->sqd.p, g->h_corr.p, g->h_sqd.p,
                       g->counters.p + 1, g->counters.p, g->mailbox, seq);
    GC(hipGetLastError());
    *seq_out = seq;
    return 0;
}

int gsicp_gicp_align(gsicp_gicp* g, const double* init, double* out) {
    Cloud &s = g->src, &t = g->tgt;
    if (s.n == 0 || t.n == 0) { g_last_error = "align: source and target must be set"; return -2; }
    hipEvent_t e0 = g->ev0, e1 = g->ev1;
    GC(hipEventRecord(e0, g->stream));
    int launches = 0;
    if (int rc = flush_points(g, s)) return rc;
    if (int rc = flush_points(g, t)) return rc;
    if (!s.cov_valid) { if (int rc = calc_cov(g, s)) return rc; ++launches; }
    if (!t.cov_valid) { if (int rc = calc_cov(g, t)) return rc; ++launches; }
    if (!g->grid_valid) { if (int rc = build_grid(g)) return rc; launches += 4; }
    const size_t ns = (size_t)(s.n_track ? s.n_track : 1);
    if (g->corr.ensure(ns) || g->sqd.ensure(ns) || g->maha.ensure(6 * ns) || g->miss.ensure(ns) || g->packed.ensure(ns) ||
        g->corr2.ensure(ns) || g->sqd2.ensure(ns) || g->maha2.ensure(6 * ns)) {
        g_last_error = "hipMalloc failed"; return -1;
    }
    AlignArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_src = s.n_track; a.src_track = s.track.p; a.src_pts = s.pts.p; a.src_cov = s.cov.p;
    a.tgt_pts = t.pts.p; a.tgt_cov = t.cov.p; a.grid = g->lv[0].view; a.fine = g->lv[1].view; a.fine_r2 = g->fine_r2;
    a.gate = g->max_corr >= 1e18 ? FLT_MAX : (float)g->max_corr * (float)g->max_corr;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) a.init[3 * r + c] = (double)(float)init[4 * r + c];
        a.init[9 + r] = (double)(float)init[4 * r + 3];
    }
    a.max_iter = g->max_iter; a.lm_max_iter = g->lm_max_iter; a.rot_eps = g->rot_eps; a.trans_eps = g->trans_eps; a.lm_init = g->lm_init;
    a.corr = g->corr.p; a.sqd = g->sqd.p; a.maha = g->maha.p; a.result = g->result.p; a.sync = g->sync.p;
    a.corr2 = g->corr2.p; a.sqd2 = g->sqd2.p; a.maha2 = g->maha2.p;
    static const bool trace_on = std::getenv("GSICP_ALIGN_TRACE") != nullptr;
    a.trace = nullptr;
    if (trace_on) { if (g->trace.ensure(512)) { g_last_error = "hipMalloc failed"; return -1; } a.trace = g->trace.p; }
    a.miss_counter = g->counters.p;
    static const int wave_prio = [] { const char* v = std::getenv("GSICP_TRACKER_WAVE_PRIO"); const int p = v ? std::atoi(v) : 0; return p < 0 ? 0 : (p > 3 ? 3 : p); }();
    a.wave_prio = wave_prio;
    a.mailbox = g->mailbox; a.seq = ++g->seq;
    int nwg = (s.n_track + AL_T - 1) / AL_T;
    if (nwg < 1) nwg = 1;
    if (nwg > AL_MAX_WG) nwg = AL_MAX_WG;
    if (nwg > g->max_resident_wg) nwg = g->max_resident_wg;   // e.g. a partitioned (CPX) device: fewer, fatter workgroups
    static const int wg_cap = [] { const char* v = std::getenv("GSICP_ALIGN_WG"); return v ? std::atoi(v) : 0; }();   // A/B switch: fewer, fatter workgroups leave the co-tenant mapper more CUs
    if (wg_cap > 0 && nwg > wg_cap) nwg = wg_cap;
    const float ms = 0.f;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (g->inject_abort && attempt == 0) {   // test hook: the first barrier of this launch sees the abort flag
            const unsigned one = 1u;
            GC(hipMemcpyAsync((char*)g->sync.p + offsetof(AlignSync, abort), &one, sizeof(one), hipMemcpyHostToDevice, g->stream));
            g->inject_abort = false;
        }
        static const int extra_launches = [] { const char* v = std::getenv("GSICP_TRACKER_EXTRA_LAUNCHES"); return v ? std::atoi(v) : 0; }();
        for (int e = 0; e < extra_launches; ++e) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, g->stream);
        { gsicp::ProfileScope ps(gsicp::ST_GICP_ALIGN, g->stream);
          hipLaunchKernelGGL(gicp_align_kernel, dim3(nwg), dim3(AL_T), 0, g->stream, a); }
        ++launches;
        GC(hipGetLastError());
        GC(hipEventRecord(e1, g->stream));
        g->dist_exact = false; g->spec_valid = false;
        if (g->spec_corr && g->spec_valid_unfetched >= 8) g->spec_corr = false;   // the caller stopped asking: stop enqueueing the export
        if (g->spec_corr && attempt == 0 && s.n_track > 0) {       // the correspondence kernels ride behind the LM kernel (no host round trip)
            ++g->spec_valid_unfetched;
            if (int rc_ = enqueue_correspondence_export(g, s.n_track, &g->spec_seq)) return rc_;
            g->spec_valid = true; g->spec_m = s.n_track;
        }
        if (int rc_ = wait_mailbox(g, &g->mailbox->align_seq, a.seq)) return rc_;
        g->host_result = g->mailbox->result;
        if (g->host_result.failed != 2) break;
        g->dist_exact = false; g->spec_valid = false;              // what rode behind a failed launch exported nothing of use
        // The grid barrier gave up: some workgroup never became resident (the GPU was saturated by a co-tenant for longer than the spin
        // budget) or the abort flag was set.  Barrier state may be stale: drain, reset, and re-run the registration as ONE workgroup —
        // grid_sum then needs no cross-workgroup barrier at all, so it cannot fail this way (slower, still entirely on the device).
        (void)hipStreamSynchronize(g->stream);
        (void)hipMemset(g->sync.p, 0, 128);
        if (nwg == 1) break;
        nwg = 1;
        a.seq = ++g->seq;
        ++g->barrier_retries;
    }
    for (int w = 0; w < 2; ++w) { g->src.staged_pending[w] = false; g->tgt.staged_pending[w] = false; }   // everything before the kernel has completed
    g->stats_pending = true;   // the events are read lazily (gsicp_gicp_last_align_stats): e1 completes a few us after the mailbox write
    if (g->host_result.failed == 2) { g_last_error = "align: the grid barrier timed out even with a single workgroup"; return -1; }
    std::memcpy(out, g->host_result.final_pose, sizeof(double) * 16);
    g->aligned = true;
    g->stats[0] = launches; g->stats[1] = g->host_result.lm_trials; g->stats[2] = g->host_result.cost;
    g->stats[3] = g->host_result.converged; g->stats[4] = ms * 1000.0; g->stats[5] = g->host_result.failed;
    return g->host_result.iterations;
}

int gsicp_gicp_get_source_correspondence(gsicp_gicp* g, int32_t* idx, float* d2, int cap) {
    if (!g->aligned) { g_last_error = "get_source_correspondence before align"; return -2; }
    const int n = g->src.n_track;
    const int m = n < cap ? n : cap;
    g->spec_corr = true;        // from now on align enqueues these kernels itself
    g->spec_valid_unfetched = 0;
    if (m > 0) {   // page-locked staging, then a plain memcpy into the caller's arrays once the export has been published
        unsigned seq = g->spec_seq;
        if (!(g->spec_valid && g->spec_m >= m)) {
            if (int rc = enqueue_correspondence_export(g, m, &seq)) return rc;
        }
        if (int rc_ = wait_mailbox(g, &g->mailbox->export_seq, seq)) return rc_;
        std::memcpy(idx, g->h_corr.p, sizeof(int) * m);
        std::memcpy(d2, g->h_sqd.p, sizeof(float) * m);
    }
    return m;
}
int gsicp_gicp_align_trace(gsicp_gicp* g, unsigned long long* out, int cap_pairs) {
    if (!g->trace.p) return 0;
    std::vector<unsigned long long> h(512);
    GC(hipStreamSynchronize(g->stream));
    GC(hipMemcpy(h.data(), g->trace.p, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int n = (int)h[511];
    if (n > cap_pairs) n = cap_pairs;
    for (int i = 0; i < 2 * n; ++i) out[i] = h[i];
    return n;
}
int gsicp_gicp_knn_stats(gsicp_gicp* g, double out[12]) {
    KnnGrid h;
    if (!g->knn_params.p) { g_last_error = "no covariance has been computed yet"; return -2; }
    GC(hipStreamSynchronize(g->stream));
    GC(hipMemcpy(&h, g->knn_params.p, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h.h; out[1] = h.nx; out[2] = h.ny; out[3] = h.nz; out[4] = h.ncells;
    for (int i = 0; i < 5; ++i) out[5 + i] = h.ring_hist[i];
    out[10] = out[11] = 0;
    return 0;
}
int gsicp_gicp_num_source(gsicp_gicp* g) { return g->src.n; }
int gsicp_gicp_num_target(gsicp_gicp* g) { return g->tgt.n; }
int gsicp_gicp_target_index_stats(gsicp_gicp* g, double out[12]) {
    for (int i = 0; i < 12; ++i) out[i] = 0.0;
    if (!g->grid_valid) return 0;
    const GridView& G = g->lv[0].view;
    out[0] = (double)G.n_sorted;
    out[1] = G.use_grid ? 1.0 : 0.0;
    for (int l = 0; l < 2; ++l) {
        const IndexLevel& L = g->lv[l];
        if (!L.view.use_grid) continue;
        unsigned cells = 0;
        GC(hipStreamSynchronize(g->stream));
        GC(hipMemcpy(&cells, L.n_cells.p, sizeof(cells), hipMemcpyDeviceToHost));       // run heads of the sorted keys = occupied coarse cells
        double* o = out + 2 + 5 * l;
        o[0] = (double)L.view.mask + 1.0;                                   // table slots
        o[1] = o[0] * 12.0 + (double)L.n_cells_host * 64.0;                 // keys + ordinals + cell records (as allocated)
        o[2] = 2.0 * (double)L.view.hf;                                     // coarse (hashed) cell edge; a fine cell is half of it
        o[3] = (double)cells;
        o[4] = (double)L.view.hf / 1.001;                                   // the radius the level is complete within
    }
    return 0;
}
int gsicp_gicp_last_align_stats(gsicp_gicp* g, double out[6]) {
    if (g->stats_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(g->ev1) == hipSuccess && hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess) g->stats[4] = ms * 1000.0;
        g->stats_pending = false;
    }
    std::memcpy(out, g->stats, sizeof(double) * 6);
    return 0;
}
int gsicp_gicp_debug_abort_next_align(gsicp_gicp* g) { g->inject_abort = true; return 0; }
int gsicp_debug_wave_sort(const float* d, const int* id, float* out_d, int* out_id, int* out_xor) {
    if (!d || !id || !out_d || !out_id || !out_xor) { g_last_error = "gsicp_debug_wave_sort: null pointer"; return -2; }
    DevBuf<float> dd, od;
    DevBuf<int> di, oi, ox;
    if (dd.ensure(64) || od.ensure(64) || di.ensure(64) || oi.ensure(64) || ox.ensure(7 * 64)) { g_last_error = "hipMalloc failed"; return -1; }
    GC(hipMemcpy(dd.p, d, 64 * sizeof(float), hipMemcpyHostToDevice));
    GC(hipMemcpy(di.p, id, 64 * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_wave_sort_kernel, dim3(1), dim3(64), 0, nullptr, dd.p, di.p, od.p, oi.p, ox.p);
    GC(hipGetLastError());
    GC(hipMemcpy(out_d, od.p, 64 * sizeof(float), hipMemcpyDeviceToHost));
    GC(hipMemcpy(out_id, oi.p, 64 * sizeof(int), hipMemcpyDeviceToHost));
    GC(hipMemcpy(out_xor, ox.p, 7 * 64 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
int gsicp_gicp_barrier_retries(gsicp_gicp* g) { return g->barrier_retries; }
int gsicp_gicp_get_final_hessian(gsicp_gicp* g, double out[36]) { std::memcpy(out, g->host_result.H_final, sizeof(double) * 36); return 0; }


// simple_knn._C.distCUDA2: out[i] = mean squared distance from point i to its 3 nearest other points (f32).  Asynchronous on `stream`.
// Scratch is kept per DEVICE (a multi-device process must not run device 1's kernels on device 0's buffers) and guarded for use from several
// streams: every call records an event behind its last kernel; a call on another stream waits for that event before it touches
// the buffers, and a call that has to grow them first waits for the device (hipFree of memory a queued kernel still reads is only safe after
// that).  The function is off the SLAM loop — the reference calls it at most when a map is created from a point cloud
// [REF scene/gaussian_model.py:20] — so the mutex simply serialises the enqueue.
namespace {
struct Knn3Scratch {
    DevBuf<float4> pts, sorted;
    DevBuf<int> cell_of;
    DevBuf<unsigned> count, start, fill;
    DevBuf<KnnGrid> params;
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
};
}  // namespace
int gsicp_knn_dist2(int P, const float* points, float* out, void* stream_v) {
    if (P < 0) { g_last_error = "gsicp_knn_dist2: negative size"; return -2; }
    if (P == 0) return 0;
    if (!points || !out) { g_last_error = "gsicp_knn_dist2: null pointer"; return -2; }
    static std::mutex mu;
    static std::map<int, Knn3Scratch>& per_device = *new std::map<int, Knn3Scratch>();   // never destroyed: no hipFree after the runtime's own teardown at exit
    std::lock_guard<std::mutex> lk(mu);
    hipStream_t stream = (hipStream_t)stream_v;
    int dev = 0;
    GC(hipGetDevice(&dev));
    Knn3Scratch& k = per_device[dev];
    if (!k.done) GC(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
    const bool grow = (size_t)P > k.pts.cap || (size_t)P > k.sorted.cap || (size_t)P > k.cell_of.cap || k.count.cap < (size_t)KNN_CELL_SLOTS || k.params.cap < (size_t)KNN_PARAM_SLOTS;
    if (k.used && grow) GC(hipDeviceSynchronize());                                     // nothing queued may still read what is about to be freed
    else if (k.used && k.last_stream != stream) GC(hipStreamWaitEvent(stream, k.done, 0));   // another stream's call owns the buffers until its kernels are done
    if (k.pts.ensure((size_t)P) || k.sorted.ensure((size_t)P) || k.cell_of.ensure((size_t)P) || k.count.ensure(KNN_CELL_SLOTS) ||
        k.start.ensure(KNN_CELL_SLOTS) || k.fill.ensure(KNN_CELL_SLOTS) || k.params.ensure(KNN_PARAM_SLOTS)) { g_last_error = "hipMalloc failed"; return -1; }
    hipLaunchKernelGGL(knn3_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, k.pts.p);
    enqueue_knn_grid_build(stream, P, k.pts.p, KNN_H_AREA, KNN_H_VOL, k.params.p, k.cell_of.p, k.count.p, k.start.p, k.fill.p, k.sorted.p);
    hipLaunchKernelGGL(knn3_grid_kernel, dim3((P + 3) / 4), dim3(256), 0, stream, P, k.pts.p, k.params.p, k.start.p, k.sorted.p, out);
    GC(hipGetLastError());
    GC(hipEventRecord(k.done, stream));
    k.last_stream = stream; k.used = true;
    return 0;
}

}  // extern "C"
