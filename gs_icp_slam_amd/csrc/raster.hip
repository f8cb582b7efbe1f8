// gfx950 rasteriser: tile binning (R2-R5), front-to-back blend (R6), back-to-front gradient replay (R7) and the
// C-ABI entry points declared in include/gsicp_hip.h.
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians / rasterize_gaussians_backward / mark_visible as called
// from [REF gaussian_renderer/__init__.py:294-302] and [REF mp_Mapper.py:242].
//
// MI355X-first design notes (details and measurements: DESIGN.md):
//  * Binning is a TWO-LEVEL stable LSD sort instead of one 64-bit sort of all duplicates: (1) the P Gaussians are
//    radix-sorted by view depth once (32-bit keys), (2) duplicates are emitted in that order with only the tile id
//    as key and radix-sorted on ceil(log2 T) bits.  Stable LSD on (depth, then tile) yields exactly the order of the
//    classic (tile<<32 | depth) sort — the parity tests check the lists bit-for-bit — while moving ~5x fewer bytes
//    through HBM/L2 (2 passes of 8 B per duplicate instead of 6 passes of 12 B).
//  * Blend kernels use one 256-thread workgroup (4 wave64s, each a 16x4 pixel strip) per 16x16 tile; splat records
//    are 48-byte packed structs staged through LDS 256 at a time and read back as wave-uniform broadcasts.
//  * blockIdx -> tile mapping is XCD-aware: block b runs on XCD b%8, so each XCD is handed a contiguous band of
//    tiles and neighbouring tiles' shared splats hit in that XCD's private L2.
//  * Backward: every lane of a wave walks the same splat at the same step, so the 10 partial gradients are reduced
//    across the 64 lanes with DPP adds (no LDS, no atomics), combined across the 4 waves in LDS, and flushed with
//    ONE global atomic per (tile, splat, component) — 256x fewer atomics than a per-pixel scheme.  Waves in which
//    no lane passes the alpha test skip the reduction entirely (wave-uniform branch).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/gsicp_hip.h"
#include "raster_common.hpp"

namespace gsicp {

thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------------ profiler
namespace {
struct ProfRec { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_log;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t g_prof_open[ST_COUNT];
std::atomic<int> g_prof_enabled{0};
const char* const g_stage_names[ST_COUNT] = {"preprocess", "depth_sort", "scan", "duplicate", "tile_sort", "tile_ranges", "blend_forward",
                                             "blend_backward", "preprocess_backward", "memset", "gicp_knn_cov", "gicp_grid_build",
                                             "gicp_align", "gicp_exact_nn"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
bool profile_on() { return g_prof_enabled.load(std::memory_order_relaxed) != 0; }
void profile_begin(int stage, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_event();
    (void)hipEventRecord(e, s);
    g_prof_open[stage] = e;
}
void profile_end(int stage, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_event();
    (void)hipEventRecord(e, s);
    g_prof_log.push_back(ProfRec{stage, g_prof_open[stage], e});
}

#define GS_CHECK(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) {                                                                                \
            g_last_error = std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" __FILE__ ":" +     \
                           std::to_string(__LINE__) + ")";                                                     \
            return -1;                                                                                         \
        }                                                                                                      \
    } while (0)

namespace {

// ------------------------------------------------------------------------------------------------ binning
struct TilesOfSorted {
    const uint32_t* tiles_touched;
    __device__ uint32_t operator()(uint32_t id) const { return tiles_touched[id]; }
};

// One thread per depth-sorted Gaussian: emit (tile id, Gaussian id) for every tile of its rectangle.
__global__ __launch_bounds__(256) void duplicate_kernel(int P, const uint32_t* __restrict__ ids_sorted,
                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles_touched,
                                                        const SplatRec* __restrict__ rec, int gx, int gy, int tile_mod, int tile_rem,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= P) return;
    const uint32_t id = ids_sorted[k];
    if (tiles_touched[id] == 0) return;
    uint32_t off = k == 0 ? 0u : offsets[k - 1];
    const SplatRec r = rec[id];
    int x0, y0, x1, y1;
    tile_rect(r.px, r.py, (int)r.radius, gx, gy, x0, y0, x1, y1);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const int t = y * gx + x;
            if (tile_mod > 1 && (t % tile_mod) != tile_rem) continue;
            keys[off] = (uint32_t)t;
            vals[off] = id;
            ++off;
        }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(int R, const uint32_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R) return;
    const uint32_t t = keys[k];
    if (k == 0 || keys[k - 1] != t) ranges[t].x = (uint32_t)k;
    if (k == R - 1 || keys[k + 1] != t) ranges[t].y = (uint32_t)(k + 1);
}

// ------------------------------------------------------------------------------------------------ blending
struct BlendArgs {
    int W, H, gx, n_tiles_local, tile_mod, tile_rem;
    const uint2* ranges;
    const uint32_t* point_list;
    const SplatRec* rec;
    const float* bg;    // device, 3 floats
    float* out_color;   // (3,H,W)
    float* out_depth;   // (H,W)
    float* final_T;     // (H,W)
    uint32_t* n_contrib;
    int* is_used;
    // backward only
    const float* dL_dpix;
    const float* dL_ddepth;
    float* dL_dmean2D;  // (P,3)
    float* dL_dconic;   // (P,4)
    float* dL_dopacity; // (P)
    float* dL_dcolors;  // (P,3)
    float* dL_ddepths;  // (P)
};

// Block b executes on XCD b % 8: give each XCD a contiguous band of tiles.
__device__ inline int xcd_band_index(int b, int nblocks) {
    const int per = nblocks >> 3;  // nblocks is a multiple of 8
    return (b & 7) * per + (b >> 3);
}

__global__ __launch_bounds__(256) void blend_forward_kernel(BlendArgs a) {
    const int tl = xcd_band_index(blockIdx.x, gridDim.x);
    if (tl >= a.n_tiles_local) return;
    const int tile = tl * a.tile_mod + a.tile_rem;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int tid = threadIdx.x;
    const int px = tx * TILE + (tid & 15), py = ty * TILE + (tid >> 4);
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint2 range = a.ranges[tile];
    int todo = (int)(range.y - range.x);

    __shared__ SplatRec s_rec[TILE_PIX];
    __shared__ uint32_t s_id[TILE_PIX];

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f;
    uint32_t contributor = 0, last_contributor = 0;

    for (uint32_t base = range.x; base < range.y; base += TILE_PIX, todo -= TILE_PIX) {
        if (__syncthreads_count(done) == TILE_PIX) break;
        const uint32_t k = base + tid;
        if (k < range.y) {
            const uint32_t id = a.point_list[k];
            s_id[tid] = id;
            s_rec[tid] = a.rec[id];
        }
        __syncthreads();
        const int n = todo < TILE_PIX ? todo : TILE_PIX;
        for (int j = 0; j < n; ++j) {
            // wave-uniform LDS broadcast reads
            const SplatRec r = s_rec[j];
            bool contrib = false;
            if (!done) {
                ++contributor;
                const float dx = r.px - pfx, dy = r.py - pfy;
                const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
                if (power <= 0.f) {
                    const float alpha = fminf(0.99f, r.opacity * __expf(power));
                    if (alpha >= 1.f / 255.f) {
                        const float test_T = T * (1.f - alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            const float w = alpha * T;
                            C0 += r.r * w; C1 += r.g * w; C2 += r.b * w; Dz += r.depth * w;
                            T = test_T;
                            last_contributor = contributor;
                            contrib = true;
                        }
                    }
                }
            }
            if (a.is_used) {
                const unsigned long long m = __ballot(contrib);
                if (m != 0ull && (tid & 63) == (__ffsll((long long)m) - 1)) a.is_used[s_id[j]] = 1;
            }
        }
    }
    if (inside) {
        const size_t HW = (size_t)a.W * a.H;
        const size_t pix = (size_t)py * a.W + px;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last_contributor;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[HW + pix] = C1 + T * a.bg[1];
        a.out_color[2 * HW + pix] = C2 + T * a.bg[2];
        a.out_depth[pix] = Dz;
    }
}

// Wave64 sum via DPP: after the call lane 63 holds the total (other lanes hold partial sums).
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
__device__ inline float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

constexpr int NGRAD = 10;  // mean2D x,y | conic a,b,c | opacity | colour r,g,b | depth

__global__ __launch_bounds__(256) void blend_backward_kernel(BlendArgs a) {
    const int tl = xcd_band_index(blockIdx.x, gridDim.x);
    if (tl >= a.n_tiles_local) return;
    const int tile = tl * a.tile_mod + a.tile_rem;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int px = tx * TILE + (tid & 15), py = ty * TILE + (tid >> 4);
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);
    const size_t HW = (size_t)a.W * a.H;
    const size_t pix = (size_t)py * a.W + px;

    __shared__ SplatRec s_rec[TILE_PIX];
    __shared__ uint32_t s_id[TILE_PIX];
    __shared__ float s_acc[TILE_PIX][NGRAD + 1];   // +1 pad: flush reads by column j stay conflict-light

    const float T_final = inside ? a.final_T[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)a.n_contrib[pix] : 0;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dpd = 0.f;
    if (inside) {
        dp0 = a.dL_dpix[pix]; dp1 = a.dL_dpix[HW + pix]; dp2 = a.dL_dpix[2 * HW + pix];
        dpd = a.dL_ddepth ? a.dL_ddepth[pix] : 0.f;
    }
    const float bg_dot = a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f;
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lcd = 0.f, last_alpha = 0.f;
    const float ddelx_dx = 0.5f * (float)a.W, ddely_dy = 0.5f * (float)a.H;

    // the tile's maximum contributor count bounds how far back we need to start
    int contributor = total;   // index (1-based) of the entry about to be processed, counted from the front
    for (int done_cnt = 0; done_cnt < total; done_cnt += TILE_PIX) {
        // stage the next 256 entries back-to-front; skip batches that lie entirely behind every pixel's last contributor
        const int n_batch = (total - done_cnt) < TILE_PIX ? (total - done_cnt) : TILE_PIX;
        if (__syncthreads_count(last_contributor > total - done_cnt - n_batch) == 0) { contributor -= n_batch; continue; }
        const int k = done_cnt + tid;
        if (k < total) {
            const uint32_t id = a.point_list[range.y - 1 - k];
            s_id[tid] = id;
            s_rec[tid] = a.rec[id];
        }
#pragma unroll
        for (int c = 0; c < NGRAD + 1; ++c) s_acc[tid][c] = 0.f;
        __syncthreads();
        const int n = n_batch;
        for (int j = 0; j < n; ++j) {
            --contributor;  // 0-based position of this entry in the tile list
            const SplatRec r = s_rec[j];
            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
            bool valid = false;
            if (contributor < last_contributor) {
                const float dx = r.px - pfx, dy = r.py - pfy;
                const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
                if (power <= 0.f) {
                    const float G = __expf(power);
                    const float alpha = fminf(0.99f, r.opacity * G);
                    if (alpha >= 1.f / 255.f) {
                        valid = true;
                        T = T / (1.f - alpha);
                        const float w = alpha * T;
                        float dL_dalpha;
                        acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = r.r;
                        acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = r.g;
                        acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = r.b;
                        accd = last_alpha * lcd + (1.f - last_alpha) * accd; lcd = r.depth;
                        dL_dalpha = (r.r - acc0) * dp0 + (r.g - acc1) * dp1 + (r.b - acc2) * dp2 + (r.depth - accd) * dpd;
                        g_r = w * dp0; g_g = w * dp1; g_b = w * dp2; g_d = w * dpd;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                        const float dL_dG = r.opacity * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * r.ca - gdy * r.cb;
                        const float dG_ddely = -gdy * r.cc - gdx * r.cb;
                        g_mx = dL_dG * dG_ddelx * ddelx_dx;
                        g_my = dL_dG * dG_ddely * ddely_dy;
                        g_ca = -0.5f * gdx * dx * dL_dG;
                        g_cb = -gdx * dy * dL_dG;
                        g_cc = -0.5f * gdy * dy * dL_dG;
                        g_op = G * dL_dalpha;
                    }
                }
            }
            if (__ballot(valid) != 0ull) {   // wave-uniform: skip splats no lane of this wave touches
                g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
                g_ca = wave_sum_to_lane63(g_ca); g_cb = wave_sum_to_lane63(g_cb); g_cc = wave_sum_to_lane63(g_cc);
                g_op = wave_sum_to_lane63(g_op);
                g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
                g_d = wave_sum_to_lane63(g_d);
                if (lane == 63) {
                    float* s = s_acc[j];
                    atomicAdd(&s[0], g_mx); atomicAdd(&s[1], g_my);
                    atomicAdd(&s[2], g_ca); atomicAdd(&s[3], g_cb); atomicAdd(&s[4], g_cc);
                    atomicAdd(&s[5], g_op);
                    atomicAdd(&s[6], g_r); atomicAdd(&s[7], g_g); atomicAdd(&s[8], g_b);
                    atomicAdd(&s[9], g_d);
                    s[NGRAD] = 1.f;   // touched flag (same value from every writer)
                }
            }
        }
        __syncthreads();
        // flush: thread j owns batch entry j -> one global atomic per (tile, splat, component)
        if (tid < n && s_acc[tid][NGRAD] != 0.f) {
            const uint32_t id = s_id[tid];
            const float* s = s_acc[tid];
            atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 0], s[0]);
            atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 1], s[1]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 0], s[2]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 1], s[3]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 2], s[4]);
            atomicAdd(&a.dL_dopacity[id], s[5]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 0], s[6]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 1], s[7]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 2], s[8]);
            atomicAdd(&a.dL_ddepths[id], s[9]);
        }
    }
}

// ================================================================================================================
// Wave-per-tile variants (default).  One wave64 owns a whole 16x16 tile: lane l handles the pixel column x = l & 15
// at the four rows y = (l >> 4) + 4k.  Consequences on CDNA4:
//   * no workgroup barriers at all (a 64-thread workgroup is one wave);
//   * the per-splat fixed cost (LDS broadcast read of the 48-byte record, loop/branch overhead, dx = px - x which is
//     shared by a lane's four pixels) is paid once per 256 pixels instead of once per 64;
//   * four independent pixel chains per lane give the in-order SIMD the ILP it needs to cover exp / LDS latency;
//   * backward: a lane first adds its four pixels' partial gradients in registers, then ONE DPP reduction per
//     (tile, splat) replaces four reductions + LDS atomics + two barriers of the 4-wave version.
// ================================================================================================================
constexpr int WPIX = 4;   // pixels per lane

__global__ __launch_bounds__(64) void blend_forward_wave_kernel(BlendArgs a) {
    const int tl = xcd_band_index(blockIdx.x, gridDim.x);
    if (tl >= a.n_tiles_local) return;
    const int tile = tl * a.tile_mod + a.tile_rem;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (lane & 15);
    const int py0 = ty * TILE + (lane >> 4);
    const float pfx = (float)px;
    const uint2 range = a.ranges[tile];

    __shared__ SplatRec s_rec[64];
    __shared__ uint32_t s_id[64];

    float T[WPIX], C0[WPIX], C1[WPIX], C2[WPIX], Dz[WPIX], pfy[WPIX];
    uint32_t last_c[WPIX];
    bool alive[WPIX], inside[WPIX];
#pragma unroll
    for (int k = 0; k < WPIX; ++k) {
        const int py = py0 + 4 * k;
        inside[k] = px < a.W && py < a.H;
        alive[k] = inside[k];
        pfy[k] = (float)py;
        T[k] = 1.f; C0[k] = C1[k] = C2[k] = Dz[k] = 0.f; last_c[k] = 0;
    }
    uint32_t pos = 0;
    for (uint32_t base = range.x; base < range.y; base += 64) {
        const bool any_alive = alive[0] | alive[1] | alive[2] | alive[3];
        if (__ballot(any_alive) == 0ull) break;
        __syncthreads();   // single-wave workgroup: orders the LDS reuse, costs ~nothing
        const uint32_t kk = base + lane;
        if (kk < range.y) {
            const uint32_t id = a.point_list[kk];
            s_id[lane] = id;
            s_rec[lane] = a.rec[id];
        }
        __syncthreads();
        const int n = (int)(range.y - base) < 64 ? (int)(range.y - base) : 64;
        for (int j = 0; j < n; ++j) {
            const SplatRec r = s_rec[j];
            ++pos;
            const float dx = r.px - pfx;
            const float adx2 = r.ca * dx * dx;
            const float bdx = r.cb * dx;
            bool contrib = false;
#pragma unroll
            for (int k = 0; k < WPIX; ++k) {
                const float dy = r.py - pfy[k];
                const float power = -0.5f * (adx2 + r.cc * dy * dy) - bdx * dy;
                const float alpha = fminf(0.99f, r.opacity * __expf(power));
                if (alive[k] && power <= 0.f && alpha >= 1.f / 255.f) {   // one (often wave-uniformly false) branch per pixel row
                    const float test_T = T[k] * (1.f - alpha);
                    if (test_T < 0.0001f) {
                        alive[k] = false;
                    } else {
                        const float w = alpha * T[k];
                        C0[k] += r.r * w; C1[k] += r.g * w; C2[k] += r.b * w; Dz[k] += r.depth * w;
                        T[k] = test_T;
                        last_c[k] = pos;
                        contrib = true;
                    }
                }
            }
            if (a.is_used) {
                const unsigned long long m = __ballot(contrib);
                if (m != 0ull && lane == (__ffsll((long long)m) - 1)) a.is_used[s_id[j]] = 1;
            }
        }
    }
    const size_t HW = (size_t)a.W * a.H;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
#pragma unroll
    for (int k = 0; k < WPIX; ++k) {
        if (inside[k]) {
            const size_t pix = (size_t)(py0 + 4 * k) * a.W + px;
            a.final_T[pix] = T[k];
            a.n_contrib[pix] = last_c[k];
            a.out_color[pix] = C0[k] + T[k] * bg0;
            a.out_color[HW + pix] = C1[k] + T[k] * bg1;
            a.out_color[2 * HW + pix] = C2[k] + T[k] * bg2;
            a.out_depth[pix] = Dz[k];
        }
    }
}

// Partial wave sum: after the call, lanes 15, 31, 47 and 63 hold the sums of their 16-lane rows (4 fused DPP adds).
__device__ inline float row_sum_to_lane15(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8
    return v;
}

__global__ __launch_bounds__(64) void blend_backward_wave_kernel(BlendArgs a) {
    const int tl = xcd_band_index(blockIdx.x, gridDim.x);
    if (tl >= a.n_tiles_local) return;
    const int tile = tl * a.tile_mod + a.tile_rem;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (lane & 15);
    const int py0 = ty * TILE + (lane >> 4);
    const float pfx = (float)px;
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);
    const size_t HW = (size_t)a.W * a.H;

    __shared__ SplatRec s_rec[64];
    __shared__ uint32_t s_id[64];
    __shared__ float s_acc[64][NGRAD + 1];

    float T[WPIX], T_final[WPIX], pfy[WPIX], dp0[WPIX], dp1[WPIX], dp2[WPIX], dpd[WPIX], bg_dot[WPIX];
    float acc0[WPIX], acc1[WPIX], acc2[WPIX], accd[WPIX], lc0[WPIX], lc1[WPIX], lc2[WPIX], lcd[WPIX], last_alpha[WPIX];
    int last_contrib[WPIX];
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    int max_contrib = 0;
#pragma unroll
    for (int k = 0; k < WPIX; ++k) {
        const int py = py0 + 4 * k;
        const bool inside = px < a.W && py < a.H;
        const size_t pix = (size_t)py * a.W + px;
        pfy[k] = (float)py;
        T_final[k] = inside ? a.final_T[pix] : 0.f;
        T[k] = T_final[k];
        last_contrib[k] = inside ? (int)a.n_contrib[pix] : 0;
        max_contrib = last_contrib[k] > max_contrib ? last_contrib[k] : max_contrib;
        dp0[k] = inside ? a.dL_dpix[pix] : 0.f;
        dp1[k] = inside ? a.dL_dpix[HW + pix] : 0.f;
        dp2[k] = inside ? a.dL_dpix[2 * HW + pix] : 0.f;
        dpd[k] = (inside && a.dL_ddepth) ? a.dL_ddepth[pix] : 0.f;
        bg_dot[k] = bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k];
        acc0[k] = acc1[k] = acc2[k] = accd[k] = 0.f;
        lc0[k] = lc1[k] = lc2[k] = lcd[k] = 0.f;
        last_alpha[k] = 0.f;
    }
    // entries at list positions >= the tile's largest n_contrib contribute to no pixel: start there
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(max_contrib, off, 64);
        max_contrib = o > max_contrib ? o : max_contrib;
    }
    const float ddelx_dx = 0.5f * (float)a.W, ddely_dy = 0.5f * (float)a.H;

    for (int top = max_contrib; top > 0; top -= 64) {   // entries [top-n, top) processed back-to-front
        const int n = top < 64 ? top : 64;
        __syncthreads();
        if (lane < n) {
            const uint32_t id = a.point_list[range.x + (uint32_t)(top - 1 - lane)];
            s_id[lane] = id;
            s_rec[lane] = a.rec[id];
        }
#pragma unroll
        for (int c = 0; c < NGRAD + 1; ++c) s_acc[lane][c] = 0.f;
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const int position = top - 1 - j;   // 0-based position in the tile list
            const SplatRec r = s_rec[j];
            const float dx = r.px - pfx;
            const float adx2 = r.ca * dx * dx;
            const float bdx = r.cb * dx;
            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
            bool valid = false;
#pragma unroll
            for (int k = 0; k < WPIX; ++k) {
                const float dy = r.py - pfy[k];
                const float power = -0.5f * (adx2 + r.cc * dy * dy) - bdx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, r.opacity * G);
                {
                    {
                        if (position < last_contrib[k] && power <= 0.f && alpha >= 1.f / 255.f) {
                            valid = true;
                            T[k] = T[k] / (1.f - alpha);
                            const float w = alpha * T[k];
                            acc0[k] = last_alpha[k] * lc0[k] + (1.f - last_alpha[k]) * acc0[k]; lc0[k] = r.r;
                            acc1[k] = last_alpha[k] * lc1[k] + (1.f - last_alpha[k]) * acc1[k]; lc1[k] = r.g;
                            acc2[k] = last_alpha[k] * lc2[k] + (1.f - last_alpha[k]) * acc2[k]; lc2[k] = r.b;
                            accd[k] = last_alpha[k] * lcd[k] + (1.f - last_alpha[k]) * accd[k]; lcd[k] = r.depth;
                            float dL_dalpha = (r.r - acc0[k]) * dp0[k] + (r.g - acc1[k]) * dp1[k] + (r.b - acc2[k]) * dp2[k] +
                                              (r.depth - accd[k]) * dpd[k];
                            g_r += w * dp0[k]; g_g += w * dp1[k]; g_b += w * dp2[k]; g_d += w * dpd[k];
                            dL_dalpha *= T[k];
                            last_alpha[k] = alpha;
                            dL_dalpha += (-T_final[k] / (1.f - alpha)) * bg_dot[k];
                            const float dL_dG = r.opacity * dL_dalpha;
                            const float gdx = G * dx, gdy = G * dy;
                            g_mx += dL_dG * (-gdx * r.ca - gdy * r.cb) * ddelx_dx;
                            g_my += dL_dG * (-gdy * r.cc - gdx * r.cb) * ddely_dy;
                            g_ca += -0.5f * gdx * dx * dL_dG;
                            g_cb += -gdx * dy * dL_dG;
                            g_cc += -0.5f * gdy * dy * dL_dG;
                            g_op += G * dL_dalpha;
                        }
                    }
                }
            }
            if (__ballot(valid) != 0ull) {   // wave-uniform: splats no pixel of the tile touches cost nothing more
                g_mx = row_sum_to_lane15(g_mx); g_my = row_sum_to_lane15(g_my);
                g_ca = row_sum_to_lane15(g_ca); g_cb = row_sum_to_lane15(g_cb); g_cc = row_sum_to_lane15(g_cc);
                g_op = row_sum_to_lane15(g_op);
                g_r = row_sum_to_lane15(g_r); g_g = row_sum_to_lane15(g_g); g_b = row_sum_to_lane15(g_b);
                g_d = row_sum_to_lane15(g_d);
                if ((lane & 15) == 15) {   // 4 row leaders combine in LDS
                    float* s = s_acc[j];
                    atomicAdd(&s[0], g_mx); atomicAdd(&s[1], g_my);
                    atomicAdd(&s[2], g_ca); atomicAdd(&s[3], g_cb); atomicAdd(&s[4], g_cc);
                    atomicAdd(&s[5], g_op);
                    atomicAdd(&s[6], g_r); atomicAdd(&s[7], g_g); atomicAdd(&s[8], g_b);
                    atomicAdd(&s[9], g_d);
                    s[NGRAD] = 1.f;
                }
            }
        }
        __syncthreads();
        if (lane < n && s_acc[lane][NGRAD] != 0.f) {
            const uint32_t id = s_id[lane];
            const float* s = s_acc[lane];
            atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 0], s[0]);
            atomicAdd(&a.dL_dmean2D[3 * (size_t)id + 1], s[1]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 0], s[2]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 1], s[3]);
            atomicAdd(&a.dL_dconic[4 * (size_t)id + 2], s[4]);
            atomicAdd(&a.dL_dopacity[id], s[5]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 0], s[6]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 1], s[7]);
            atomicAdd(&a.dL_dcolors[3 * (size_t)id + 2], s[8]);
            atomicAdd(&a.dL_ddepths[id], s[9]);
        }
    }
    (void)total;
}

inline int blend_variant() {   // GSICP_BLEND=block selects the 4-wave-per-tile kernels (kept for A/B measurements)
    static const int v = [] { const char* e = getenv("GSICP_BLEND"); return (e && std::strcmp(e, "block") == 0) ? 1 : 0; }();
    return v;
}

inline int tile_bits(int T) {
    int b = 1;
    while ((1 << b) < T) ++b;
    return b;
}

inline size_t geom_temp_bytes(int P, hipStream_t s) {
    size_t sort_b = 0, scan_b = 0;
    const int n = P > 0 ? P : 1;
    (void)rocprim::radix_sort_pairs(nullptr, sort_b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                              (size_t)n, 0, 32, s);
    auto it = rocprim::make_transform_iterator((const uint32_t*)nullptr, TilesOfSorted{nullptr});
    (void)rocprim::inclusive_scan(nullptr, scan_b, it, (uint32_t*)nullptr, (size_t)n, rocprim::plus<uint32_t>(), s);
    return sort_b > scan_b ? sort_b : scan_b;
}
inline size_t bin_temp_bytes(size_t R, int bits, hipStream_t s) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                              R > 0 ? R : 1, 0, bits, s);
    return b;
}

}  // namespace
}  // namespace gsicp

using namespace gsicp;

extern "C" {

int gsicp_abi_version(void) { return GSICP_ABI_VERSION; }
const char* gsicp_last_error(void) { return g_last_error.c_str(); }
int gsicp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int gsicp_profile_enable(int on) { g_prof_enabled.store(on ? 1 : 0); return 0; }
int gsicp_profile_num_stages(void) { return ST_COUNT; }
const char* gsicp_profile_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? g_stage_names[stage] : ""; }
int gsicp_profile_read(double* ms_out, int* count_out, int capacity) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < capacity && i < ST_COUNT; ++i) { ms_out[i] = 0.0; if (count_out) count_out[i] = 0; }
    for (const ProfRec& r : g_prof_log) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (r.stage < capacity) { ms_out[r.stage] += ms; if (count_out) count_out[r.stage] += 1; }
        g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b);
    }
    g_prof_log.clear();
    return ST_COUNT;
}

int gsicp_raster_layout(int P, int num_rendered, int width, int height, size_t out[10]) {
    const int T = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    const GeomLayout G = geom_layout(P, geom_temp_bytes(P, nullptr));
    const BinLayout B = bin_layout((size_t)num_rendered, bin_temp_bytes((size_t)num_rendered, tile_bits(T), nullptr));
    const ImgLayout I = img_layout(width, height);
    out[0] = G.total; out[1] = B.total; out[2] = I.total; out[3] = G.records; out[4] = B.point_list; out[5] = B.tile_keys;
    out[6] = I.ranges; out[7] = I.final_T; out[8] = I.n_contrib; out[9] = G.clamped;
    return 0;
}

int gsicp_raster_forward(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                         gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                         int height, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                         int* is_used, int tile_mod, int tile_rem, int debug, void* stream_v) {
    (void)prefiltered; (void)debug;
    hipStream_t stream = (hipStream_t)stream_v;
    if (width <= 0 || height <= 0 || P < 0) { g_last_error = "gsicp_raster_forward: bad sizes"; return -2; }
    if (P > 0 && (shs == nullptr) == (colors_precomp == nullptr)) { g_last_error = "provide exactly one of shs / colors_precomp"; return -2; }
    if (P > 0 && ((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) {
        g_last_error = "provide exactly one of (scales, rotations) / cov3D_precomp"; return -2;
    }
    if (shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1))) { g_last_error = "sh degree / coefficient count mismatch"; return -2; }
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "bad tile_mod / tile_rem"; return -2; }
    if (!geom_alloc || !binning_alloc || !img_alloc) { g_last_error = "null resize callback"; return -2; }

    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, T = gx * gy;
    const size_t HW = (size_t)width * height;

    const ImgLayout IL = img_layout(width, height);
    char* img = img_alloc(img_user, IL.total);
    if (!img) { g_last_error = "img resize callback returned NULL"; return -3; }
    uint2* ranges = (uint2*)(img + IL.ranges);
    float* final_T = (float*)(img + IL.final_T);
    uint32_t* n_contrib = (uint32_t*)(img + IL.n_contrib);

    GS_CHECK(hipMemsetAsync(ranges, 0, (size_t)T * 8, stream));
    if (is_used && P > 0) GS_CHECK(hipMemsetAsync(is_used, 0, (size_t)P * 4, stream));

    int num_rendered = 0;
    const GeomLayout GL = geom_layout(P, geom_temp_bytes(P, stream));
    char* geom = geom_alloc(geom_user, GL.total);
    if (!geom) { g_last_error = "geom resize callback returned NULL"; return -3; }
    SplatRec* rec = (SplatRec*)(geom + GL.records);
    uint32_t* point_list = nullptr;

    if (P > 0) {
        PreprocessArgs pa;
        pa.P = P; pa.D = D; pa.M = M; pa.W = width; pa.H = height;
        pa.means3D = means3D; pa.shs = shs; pa.colors_precomp = colors_precomp; pa.opacities = opacities; pa.scales = scales;
        pa.rotations = rotations; pa.cov3D_precomp = cov3D_precomp; pa.scale_modifier = scale_modifier;
        pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos; pa.tanfovx = tan_fovx; pa.tanfovy = tan_fovy;
        pa.tile_mod = tile_mod; pa.tile_rem = tile_rem;
        pa.rec = rec; pa.clamped = (unsigned char*)(geom + GL.clamped);
        pa.tiles_touched = (uint32_t*)(geom + GL.tiles_touched);
        pa.depth_keys = (uint32_t*)(geom + GL.depth_keys); pa.ids = (uint32_t*)(geom + GL.ids);
        pa.radii = radii;
        { ProfileScope ps(ST_PREPROCESS, stream); launch_preprocess(pa, stream); }

        uint32_t* keys_sorted = (uint32_t*)(geom + GL.depth_keys_sorted);
        uint32_t* ids_sorted = (uint32_t*)(geom + GL.ids_sorted);
        uint32_t* offsets = (uint32_t*)(geom + GL.offsets);
        size_t tb = GL.temp_bytes;
        { ProfileScope ps(ST_DEPTH_SORT, stream);
          GS_CHECK(rocprim::radix_sort_pairs(geom + GL.temp, tb, pa.depth_keys, keys_sorted, pa.ids, ids_sorted, (size_t)P, 0, 32, stream)); }
        auto it = rocprim::make_transform_iterator((const uint32_t*)ids_sorted, TilesOfSorted{pa.tiles_touched});
        tb = GL.temp_bytes;
        { ProfileScope ps(ST_SCAN, stream);
          GS_CHECK(rocprim::inclusive_scan(geom + GL.temp, tb, it, offsets, (size_t)P, rocprim::plus<uint32_t>(), stream)); }
        uint32_t total = 0;
        GS_CHECK(hipMemcpyAsync(&total, offsets + (P - 1), 4, hipMemcpyDeviceToHost, stream));
        GS_CHECK(hipStreamSynchronize(stream));
        num_rendered = (int)total;

        const int bits = tile_bits(T);
        const BinLayout BL = bin_layout((size_t)num_rendered, bin_temp_bytes((size_t)num_rendered, bits, stream));
        char* bin = binning_alloc(binning_user, BL.total);
        if (!bin) { g_last_error = "binning resize callback returned NULL"; return -3; }
        point_list = (uint32_t*)(bin + BL.point_list);
        if (num_rendered > 0) {
            uint32_t* keys_u = (uint32_t*)(bin + BL.tile_keys_unsorted);
            uint32_t* vals_u = (uint32_t*)(bin + BL.point_list_unsorted);
            uint32_t* keys_s = (uint32_t*)(bin + BL.tile_keys);
            { ProfileScope ps(ST_DUPLICATE, stream);
              hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, ids_sorted, offsets,
                                 pa.tiles_touched, rec, gx, gy, tile_mod, tile_rem, keys_u, vals_u); }
            size_t tb2 = BL.temp_bytes;
            { ProfileScope ps(ST_TILE_SORT, stream);
              GS_CHECK(rocprim::radix_sort_pairs(bin + BL.temp, tb2, keys_u, keys_s, vals_u, point_list, (size_t)num_rendered, 0, bits, stream)); }
            { ProfileScope ps(ST_RANGES, stream);
              hipLaunchKernelGGL(tile_ranges_kernel, dim3((num_rendered + 255) / 256), dim3(256), 0, stream, num_rendered, keys_s, ranges); }
        }
    } else {
        GS_CHECK(hipStreamSynchronize(stream));
        char* bin = binning_alloc(binning_user, bin_layout(0, 0).total);
        if (!bin) { g_last_error = "binning resize callback returned NULL"; return -3; }
        point_list = (uint32_t*)bin;
    }

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem;
    ba.n_tiles_local = (T - tile_rem + tile_mod - 1) / tile_mod;
    ba.ranges = ranges; ba.point_list = point_list; ba.rec = rec;
    ba.bg = background;
    ba.out_color = out_color; ba.out_depth = out_depth; ba.final_T = final_T; ba.n_contrib = n_contrib; ba.is_used = is_used;
    (void)HW;
    if (ba.n_tiles_local > 0) {
        const int nblocks = (ba.n_tiles_local + 7) / 8 * 8;
        ProfileScope ps(ST_BLEND_FWD, stream);
        if (blend_variant() == 1) hipLaunchKernelGGL(blend_forward_kernel, dim3(nblocks), dim3(TILE_PIX), 0, stream, ba);
        else hipLaunchKernelGGL(blend_forward_wave_kernel, dim3(nblocks), dim3(64), 0, stream, ba);
    }
    GS_CHECK(hipGetLastError());
    return num_rendered;
}

int gsicp_raster_backward(int P, int D, int M, int num_rendered, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                          const char* geom_buffer, const char* binning_buffer, const char* img_buffer, const float* dL_dpix,
                          const float* dL_ddepth, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                          float* dL_ddepths, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                          float* dL_drots, int tile_mod, int tile_rem, int debug, void* stream_v) {
    (void)debug;
    hipStream_t stream = (hipStream_t)stream_v;
    if (P <= 0) return 0;
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "bad tile_mod / tile_rem"; return -2; }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, T = gx * gy;
    const GeomLayout GL = geom_layout(P, 0);
    const BinLayout BL = bin_layout((size_t)num_rendered, 0);
    const ImgLayout IL = img_layout(width, height);

    const bool prof_ms = profile_on();
    if (prof_ms) profile_begin(ST_MEMSET, stream);
    GS_CHECK(hipMemsetAsync(dL_dmeans2D, 0, (size_t)P * 12, stream));
    GS_CHECK(hipMemsetAsync(dL_dconic, 0, (size_t)P * 16, stream));
    GS_CHECK(hipMemsetAsync(dL_dopacity, 0, (size_t)P * 4, stream));
    GS_CHECK(hipMemsetAsync(dL_dcolors, 0, (size_t)P * 12, stream));
    GS_CHECK(hipMemsetAsync(dL_ddepths, 0, (size_t)P * 4, stream));
    if (prof_ms) profile_end(ST_MEMSET, stream);

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem;
    ba.n_tiles_local = (T - tile_rem + tile_mod - 1) / tile_mod;
    ba.ranges = (const uint2*)(img_buffer + IL.ranges);
    ba.point_list = (const uint32_t*)(binning_buffer + BL.point_list);
    ba.rec = (const SplatRec*)(geom_buffer + GL.records);
    ba.bg = background;
    ba.final_T = (float*)(img_buffer + IL.final_T);
    ba.n_contrib = (uint32_t*)(img_buffer + IL.n_contrib);
    ba.dL_dpix = dL_dpix; ba.dL_ddepth = dL_ddepth;
    ba.dL_dmean2D = dL_dmeans2D; ba.dL_dconic = dL_dconic; ba.dL_dopacity = dL_dopacity; ba.dL_dcolors = dL_dcolors;
    ba.dL_ddepths = dL_ddepths;
    if (num_rendered > 0 && ba.n_tiles_local > 0) {
        const int nblocks = (ba.n_tiles_local + 7) / 8 * 8;
        ProfileScope ps(ST_BLEND_BWD, stream);
        if (blend_variant() == 1) hipLaunchKernelGGL(blend_backward_kernel, dim3(nblocks), dim3(TILE_PIX), 0, stream, ba);
        else hipLaunchKernelGGL(blend_backward_wave_kernel, dim3(nblocks), dim3(64), 0, stream, ba);
    }

    PreprocessBwdArgs pb;
    pb.P = P; pb.D = D; pb.M = M; pb.W = width; pb.H = height;
    pb.means3D = means3D; pb.shs = shs; pb.colors_precomp = colors_precomp; pb.scales = scales; pb.rotations = rotations;
    pb.cov3D_precomp = cov3D_precomp; pb.scale_modifier = scale_modifier; pb.view = viewmatrix; pb.proj = projmatrix;
    pb.campos = cam_pos; pb.tanfovx = tan_fovx; pb.tanfovy = tan_fovy; pb.radii = radii;
    pb.clamped = (const unsigned char*)(geom_buffer + GL.clamped);
    pb.dL_dmean2D = dL_dmeans2D; pb.dL_dconic = dL_dconic; pb.dL_dcolors = dL_dcolors; pb.dL_ddepths = dL_ddepths;
    pb.dL_dmeans3D = dL_dmeans3D; pb.dL_dcov3D = dL_dcov3D; pb.dL_dsh = dL_dsh; pb.dL_dscales = dL_dscales; pb.dL_drots = dL_drots;
    { ProfileScope ps(ST_PREPROCESS_BWD, stream); launch_preprocess_backward(pb, stream); }
    GS_CHECK(hipGetLastError());
    return 0;
}

int gsicp_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                              unsigned char* present, void* stream) {
    (void)projmatrix;
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    GS_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
