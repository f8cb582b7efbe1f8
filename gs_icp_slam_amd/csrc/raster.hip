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
//  * Blend kernels use one 256-thread workgroup (4 wave64s, each a 16x4 pixel strip) per 16x16 tile.  Binning tags
//    every list entry with 4 strip bits (which strips the splat's alpha >= 1/255 footprint can reach); each wave
//    compacts its own 64-entry batches by those bits and stages only the surviving 48-byte records in a
//    wave-private LDS slab — no workgroup barriers, and culled entries cost 1/64 of a vector instruction.
//  * blockIdx -> tile mapping is XCD-aware: block b runs on XCD b%8, so each XCD is handed a contiguous band of
//    tiles and neighbouring tiles' shared splats hit in that XCD's private L2.
//  * Backward: every lane of a wave walks the same splat at the same step, so the 10 partial gradients are reduced
//    across the 64 lanes with 60 hand-scheduled DPP adds (no LDS atomics, no wait states), parked in the wave's LDS
//    slab and flushed with one 64-wide global atomic instruction per component per batch — 64x fewer atomics than
//    a per-pixel scheme.  Waves in which no lane passes the alpha test skip the reduction (wave-uniform branch).
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
                                             "blend_backward", "preprocess_backward", "entry_grad_sum", "gicp_knn_cov", "gicp_grid_build",
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

// One thread per depth-sorted Gaussian: emit (tile id, Gaussian id | strip bits) for every tile of its rectangle.
__global__ __launch_bounds__(256) void duplicate_kernel(int P, const uint32_t* __restrict__ ids_sorted,
                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles_touched,
                                                        const SplatRec* __restrict__ rec, const int* __restrict__ radii, int gx, int gy,
                                                        int tile_mod, int tile_rem, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t* __restrict__ entry_gauss) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= P) return;
    const uint32_t id = ids_sorted[k];
    if (tiles_touched[id] == 0) return;
    uint32_t off = k == 0 ? 0u : offsets[k - 1];
    const SplatRec r = rec[id];
    int x0, y0, x1, y1;
    tile_rect(r.px, r.py, radii[id], gx, gy, x0, y0, x1, y1);
    const float fx0 = r.px - r.hx, fx1 = r.px + r.hx, fy0 = r.py - r.hy, fy1 = r.py + r.hy;   // alpha footprint box
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const int t = y * gx + x;
            if (tile_mod > 1 && (t % tile_mod) != tile_rem) continue;
            uint32_t bits = 0;
            if (fx1 >= (float)(x * TILE) && fx0 <= (float)(x * TILE + TILE - 1)) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const float ylo = (float)(y * TILE + 4 * sidx);
                    if (fy1 >= ylo && fy0 <= ylo + 3.f) bits |= 1u << sidx;
                }
            }
            keys[off] = (uint32_t)t;
            vals[off] = off | (bits << STRIP_SHIFT);   // the list carries the emission slot; the slot knows its Gaussian
            entry_gauss[off] = id;
            ++off;
        }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(int R, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                          uint2* __restrict__ ranges, uint32_t* __restrict__ entry_pos) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R) return;
    entry_pos[vals[k] & ID_MASK] = (uint32_t)k;   // inverse of the tile sort, used by the per-Gaussian gradient gather
    const uint32_t t = keys[k];
    if (k == 0 || keys[k - 1] != t) ranges[t].x = (uint32_t)k;
    if (k == R - 1 || keys[k + 1] != t) ranges[t].y = (uint32_t)(k + 1);
}

// Longest-processing-time-first dispatch: key = ~length so that an ascending radix sort yields longest tiles first.
__global__ __launch_bounds__(256) void tile_order_keys_kernel(int n_local, int tile_mod, int tile_rem, const uint2* __restrict__ ranges,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_local) return;
    const int t = i * tile_mod + tile_rem;
    const uint2 r = ranges[t];
    keys[i] = 0xFFFFFFFFu - (r.y - r.x);
    vals[i] = (uint32_t)t;
}

// ------------------------------------------------------------------------------------------------ blending
struct BlendArgs {
    int W, H, gx, n_tiles_local, tile_mod, tile_rem;
    const uint2* ranges;
    const uint32_t* order;     // local tiles, longest list first
    const uint32_t* point_list;
    const SplatRec* rec;
    const float* bg;    // device, 3 floats
    float* out_color;   // (3,H,W)
    float* out_depth;   // (H,W)
    float* final_T;     // (H,W)
    uint32_t* n_contrib;
    int* is_used;
    const uint32_t* entry_gauss;   // emission slot -> Gaussian id
    // backward only
    const float* dL_dpix;
    const float* dL_ddepth;
    float* slots;        // (R, 4, SLOT_F) per-(emission slot, strip) gradient sums
};

// ================================================================================================================
// "Strip" blend kernels (default).  A 256-thread workgroup owns a 16x16 tile; wave s owns the 16x4 strip of rows
// 4s..4s+3 and runs on its own — no workgroup barrier anywhere.  Per 64 list entries a wave
//   1. loads the entries with one coalesced vector load and keeps those whose strip bit (computed at binning time
//      from the Gaussian's alpha >= 1/255 footprint) is set — typically a minority, since a Gaussian of ~10 px
//      radius touches one or two of a tile's four strips;
//   2. compacts the survivors with ballot/mbcnt, gathers their 48-byte records with all lanes in parallel into a
//      wave-private LDS slab (the gather latency is paid once per batch, not once per entry);
//   3. walks the compacted slab with wave-uniform LDS broadcast reads.
// Entries that cannot touch the strip therefore cost 1/64 of a vector instruction instead of a full evaluation, and
// the four waves never wait for each other.  Exactness is untouched: the per-pixel alpha / transmittance tests are
// still what decides, the strip bit only removes entries every pixel of the strip would have rejected anyway.
// ================================================================================================================
constexpr int SLAB = 64;

__global__ __launch_bounds__(64) void blend_forward_strip_kernel(BlendArgs a) {
    // one 64-thread workgroup per (tile, strip), dispatched longest tile first: the hardware hands workgroups to CUs
    // in index order, so sorting by list length is LPT scheduling and the kernel no longer ends on a few stragglers
    const int tl = (int)(blockIdx.x >> 2);
    if (tl >= a.n_tiles_local) return;
    const int wave = (int)(blockIdx.x & 3);
    const int tile = (int)a.order[tl];
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (lane & 15), py = ty * TILE + wave * 4 + (lane >> 4);
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint2 range = a.ranges[tile];
    const uint32_t strip_bit = 1u << (STRIP_SHIFT + wave);

    __shared__ SplatRec s_rec[SLAB];
    __shared__ uint32_t s_id[SLAB];
    __shared__ uint32_t s_pos[SLAB];

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dz = 0.f;
    uint32_t last_contributor = 0;

    for (uint32_t base = range.x; base < range.y; base += 64) {
        if (__ballot(!done) == 0ull) break;   // every pixel of the strip has saturated
        const uint32_t k = base + lane;
        uint32_t e = 0;
        if (k < range.y) e = a.point_list[k];
        const bool keep = (e & strip_bit) != 0;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) continue;
        const int n = __popcll(m);
        if (keep) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            const uint32_t id = a.entry_gauss[e & ID_MASK];
            s_id[slot] = id;
            s_pos[slot] = k - range.x + 1;   // 1-based position in the tile list (the reference's "contributor")
            s_rec[slot] = a.rec[id];
        }
        __builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in order; this only stops compiler reordering
        SplatRec rn = s_rec[0];
        uint32_t pn = s_pos[0];
        for (int j = 0; j < n; ++j) {
            const SplatRec r = rn;
            const uint32_t pos_j = pn;
            const int jn = j + 1 < n ? j + 1 : j;
            rn = s_rec[jn];          // LDS reads for the next entry are in flight while this one is evaluated
            pn = s_pos[jn];
            const float dx = r.px - pfx, dy = r.py - pfy;
            const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
            const float alpha = fminf(0.99f, r.opacity * __expf(power));
            bool contrib = false;
            if (!done && power <= 0.f && alpha >= 1.f / 255.f) {
                const float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) {
                    done = true;
                } else {
                    const float w = alpha * T;
                    C0 += r.r * w; C1 += r.g * w; C2 += r.b * w; Dz += r.depth * w;
                    T = test_T;
                    last_contributor = pos_j;
                    contrib = true;
                }
            }
            if (a.is_used) {
                const unsigned long long mc = __ballot(contrib);
                if (mc != 0ull && lane == (__ffsll((long long)mc) - 1)) a.is_used[s_id[j]] = 1;
            }
        }
        __builtin_amdgcn_wave_barrier();
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

// Wave64 sums of ten values at once: 6 DPP steps x 10 values = exactly 60 VALU instructions.  Doing the steps
// value-interleaved (all ten values per step) keeps every DPP source >= 10 instructions behind its producer, so no
// wait states are needed, and writing it by hand avoids the v_mov / s_nop padding hipcc emits around
// partially-masked row_bcast moves.  Disabled lanes / rows keep their old value (dst == src).  Lane 63 ends up
// with the wave totals.
#define GS_DPP10(CTRL)                                                                              \
    asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\t"     \
                 "v_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"     \
                 "v_add_f32_dpp %4, %4, %4 " CTRL "\n\tv_add_f32_dpp %5, %5, %5 " CTRL "\n\t"     \
                 "v_add_f32_dpp %6, %6, %6 " CTRL "\n\tv_add_f32_dpp %7, %7, %7 " CTRL "\n\t"     \
                 "v_add_f32_dpp %8, %8, %8 " CTRL "\n\tv_add_f32_dpp %9, %9, %9 " CTRL            \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9))
__device__ inline void wave_sum10(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6, float& v7,
                                  float& v8, float& v9) {
    asm volatile("s_nop 1");   // the first DPP reads registers the preceding VALU code may just have written
    GS_DPP10("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    GS_DPP10("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    GS_DPP10("row_shr:4 row_mask:0xf bank_mask:0xf");
    GS_DPP10("row_shr:8 row_mask:0xf bank_mask:0xf");
    GS_DPP10("row_bcast:15 row_mask:0xa bank_mask:0xf");
    GS_DPP10("row_bcast:31 row_mask:0xc bank_mask:0xf");
}

__global__ __launch_bounds__(64) void blend_backward_strip_kernel(BlendArgs a) {
    // one 64-thread workgroup per (tile, strip), dispatched longest tile first: the hardware hands workgroups to CUs
    // in index order, so sorting by list length is LPT scheduling and the kernel no longer ends on a few stragglers
    const int tl = (int)(blockIdx.x >> 2);
    if (tl >= a.n_tiles_local) return;
    const int wave = (int)(blockIdx.x & 3);
    const int tile = (int)a.order[tl];
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x;
    const int px = tx * TILE + (lane & 15), py = ty * TILE + wave * 4 + (lane >> 4);
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint2 range = a.ranges[tile];
    const uint32_t strip_bit = 1u << (STRIP_SHIFT + wave);
    const size_t HW = (size_t)a.W * a.H;
    const size_t pix = (size_t)py * a.W + px;

    __shared__ SplatRec s_rec[SLAB];
    __shared__ int s_pos[SLAB];
    __shared__ uint32_t s_u[SLAB];
    __shared__ float s_sum[SLAB][NGRAD + 1];

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

    // entries at positions >= the strip's largest n_contrib reach no pixel of this strip: their slots only get zeros
    int top0 = last_contributor;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(top0, off, 64);
        top0 = o > top0 ? o : top0;
    }
    const int total = (int)(range.y - range.x);
    for (int top = total; top > 0; top -= 64) {   // entries [top-64, top) back-to-front; lane 0 holds the last one
        const int posn = top - 1 - lane;   // 0-based list position of this lane's entry
        uint32_t e = 0;
        if (posn >= 0) e = a.point_list[range.x + (uint32_t)posn];
        const bool keep = (e & strip_bit) != 0;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) continue;
        if (top - 64 >= top0) {   // whole batch lies behind every pixel's last contributor: zero slots, nothing to compute
            if (keep) {
                float4* dst = (float4*)(a.slots + ((size_t)(e & ID_MASK) * 4 + wave) * SLOT_F);
                dst[0] = make_float4(0.f, 0.f, 0.f, 0.f); dst[1] = dst[0]; dst[2] = dst[0];
            }
            continue;
        }
        const int n = __popcll(m);
        if (keep) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            s_pos[slot] = posn;
            s_u[slot] = e & ID_MASK;
            s_rec[slot] = a.rec[a.entry_gauss[e & ID_MASK]];
        }
#pragma unroll
        for (int c = 0; c < NGRAD; ++c) s_sum[lane][c] = 0.f;
        __builtin_amdgcn_wave_barrier();
        SplatRec rn = s_rec[0];
        int pn = s_pos[0];
        for (int j = 0; j < n; ++j) {
            const SplatRec r = rn;
            const int position = pn;
            const int jn = j + 1 < n ? j + 1 : j;
            rn = s_rec[jn];          // prefetch the next entry's record
            pn = s_pos[jn];
            const float dx = r.px - pfx, dy = r.py - pfy;
            const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, r.opacity * G);
            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
            const bool valid = position < last_contributor && power <= 0.f && alpha >= 1.f / 255.f;
            if (valid) {
                const float inv_one_m = __builtin_amdgcn_rcpf(1.f - alpha);   // alpha <= 0.99: one v_rcp_f32 (1 ulp) instead of two IEEE divisions
                T = T * inv_one_m;
                const float w = alpha * T;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = r.r;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = r.g;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = r.b;
                accd = last_alpha * lcd + (1.f - last_alpha) * accd; lcd = r.depth;
                float dL_dalpha = (r.r - acc0) * dp0 + (r.g - acc1) * dp1 + (r.b - acc2) * dp2 + (r.depth - accd) * dpd;
                g_r = w * dp0; g_g = w * dp1; g_b = w * dp2; g_d = w * dpd;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv_one_m) * bg_dot;
                const float dL_dG = r.opacity * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                g_mx = dL_dG * (-gdx * r.ca - gdy * r.cb) * ddelx_dx;
                g_my = dL_dG * (-gdy * r.cc - gdx * r.cb) * ddely_dy;
                g_ca = -0.5f * gdx * dx * dL_dG;
                g_cb = -gdx * dy * dL_dG;
                g_cc = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            if (__ballot(valid) != 0ull) {   // wave-uniform
                wave_sum10(g_mx, g_my, g_ca, g_cb, g_cc, g_op, g_r, g_g, g_b, g_d);
                if (lane == 63) {
                    float* sp = s_sum[j];
                    sp[0] = g_mx; sp[1] = g_my; sp[2] = g_ca; sp[3] = g_cb; sp[4] = g_cc; sp[5] = g_op;
                    sp[6] = g_r; sp[7] = g_g; sp[8] = g_b; sp[9] = g_d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // flush this wave's slab: lane j owns compacted entry j and stores its sums (zeros included) into the
        // entry's private slot for this strip — plain 16-byte stores, no atomics, bit-reproducible gradients
        if (lane < n) {
            const float* sp = s_sum[lane];
            float4* dst = (float4*)(a.slots + ((size_t)s_u[lane] * 4 + wave) * SLOT_F);   // slots are indexed by emission slot
            dst[0] = make_float4(sp[0], sp[1], sp[2], sp[3]);
            dst[1] = make_float4(sp[4], sp[5], sp[6], sp[7]);
            dst[2] = make_float4(sp[8], sp[9], 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Entry-parallel: add up each emission slot's (up to four) strip slots into one 12-float record, so that the
// per-Gaussian pass only streams a contiguous array.  Fully parallel, no dependent chains.
__global__ __launch_bounds__(256) void entry_sum_kernel(int R, const uint32_t* __restrict__ entry_bits, const float* __restrict__ slots,
                                                        float* __restrict__ entry_sum) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= R) return;
    const uint32_t bits = entry_bits[u] >> STRIP_SHIFT;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        if ((bits >> sidx) & 1u) {
            const float4* sl = (const float4*)(slots + ((size_t)u * 4 + sidx) * SLOT_F);
            const float4 v0 = sl[0], v1 = sl[1], v2 = sl[2];
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y;
        }
    }
    float4* dst = (float4*)(entry_sum + (size_t)u * SLOT_F);
    dst[0] = a0; dst[1] = a1; dst[2] = a2;
}

// One thread per Gaussian IN DEPTH-RANK ORDER: rank r owns the contiguous emission slots [offsets[r-1], offsets[r]), so a
// wave reads one contiguous stretch of entry_sum (coalesced, TLB-friendly) and scatters ten sums per Gaussian.
__global__ __launch_bounds__(256) void gaussian_grad_gather_kernel(int P, const uint32_t* __restrict__ ids_sorted,
                                                                   const uint32_t* __restrict__ offsets, const float* __restrict__ entry_sum,
                                                                   float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                                                   float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
                                                                   float* __restrict__ dL_ddepths) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= P) return;
    const uint32_t id = ids_sorted[r];
    const uint32_t u0 = r == 0 ? 0u : offsets[r - 1], u1 = offsets[r];
    float gs[NGRAD];
#pragma unroll
    for (int c = 0; c < NGRAD; ++c) gs[c] = 0.f;
    const float4* es = (const float4*)entry_sum;
    for (uint32_t u = u0; u < u1; ++u) {
        const float4 p0 = es[3 * (size_t)u], p1 = es[3 * (size_t)u + 1], p2 = es[3 * (size_t)u + 2];
        gs[0] += p0.x; gs[1] += p0.y; gs[2] += p0.z; gs[3] += p0.w;
        gs[4] += p1.x; gs[5] += p1.y; gs[6] += p1.z; gs[7] += p1.w;
        gs[8] += p2.x; gs[9] += p2.y;
    }
    dL_dmean2D[3 * (size_t)id] = gs[0]; dL_dmean2D[3 * (size_t)id + 1] = gs[1]; dL_dmean2D[3 * (size_t)id + 2] = 0.f;
    dL_dconic[4 * (size_t)id] = gs[2]; dL_dconic[4 * (size_t)id + 1] = gs[3]; dL_dconic[4 * (size_t)id + 2] = gs[4]; dL_dconic[4 * (size_t)id + 3] = 0.f;
    dL_dopacity[id] = gs[5];
    dL_dcolors[3 * (size_t)id] = gs[6]; dL_dcolors[3 * (size_t)id + 1] = gs[7]; dL_dcolors[3 * (size_t)id + 2] = gs[8];
    dL_ddepths[id] = gs[9];
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
inline size_t order_temp_bytes(int T, hipStream_t s) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (size_t)(T > 0 ? T : 1), 0, 32, s);
    return b;
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

int gsicp_raster_layout(int P, int num_rendered, int width, int height, size_t out[12]) {
    const int T = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    const GeomLayout G = geom_layout(P, geom_temp_bytes(P, nullptr));
    const BinLayout B = bin_layout((size_t)num_rendered, bin_temp_bytes((size_t)num_rendered, tile_bits(T), nullptr));
    const ImgLayout I = img_layout(width, height, order_temp_bytes(T, nullptr));
    out[0] = G.total; out[1] = B.total; out[2] = I.total; out[3] = G.records; out[4] = B.point_list; out[5] = B.tile_keys;
    out[6] = I.ranges; out[7] = I.final_T; out[8] = I.n_contrib; out[9] = G.clamped; out[10] = B.entry_gauss; out[11] = B.entry_pos;
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

    const ImgLayout IL = img_layout(width, height, order_temp_bytes(T, stream));
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
    uint32_t* entry_gauss = nullptr;

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
        entry_gauss = (uint32_t*)(bin + BL.entry_gauss);
        if (num_rendered > 0) {
            uint32_t* keys_u = (uint32_t*)(bin + BL.tile_keys_unsorted);
            uint32_t* vals_u = (uint32_t*)(bin + BL.point_list_unsorted);
            uint32_t* keys_s = (uint32_t*)(bin + BL.tile_keys);
            { ProfileScope ps(ST_DUPLICATE, stream);
              hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, ids_sorted, offsets,
                                 pa.tiles_touched, rec, radii, gx, gy, tile_mod, tile_rem, keys_u, vals_u, (uint32_t*)(bin + BL.entry_gauss)); }
            size_t tb2 = BL.temp_bytes;
            { ProfileScope ps(ST_TILE_SORT, stream);
              GS_CHECK(rocprim::radix_sort_pairs(bin + BL.temp, tb2, keys_u, keys_s, vals_u, point_list, (size_t)num_rendered, 0, bits, stream)); }
            { ProfileScope ps(ST_RANGES, stream);
              hipLaunchKernelGGL(tile_ranges_kernel, dim3((num_rendered + 255) / 256), dim3(256), 0, stream, num_rendered, keys_s, point_list, ranges,
                                 (uint32_t*)(bin + BL.entry_pos)); }
        }
    } else {
        GS_CHECK(hipStreamSynchronize(stream));
        char* bin = binning_alloc(binning_user, bin_layout(0, 0).total);
        if (!bin) { g_last_error = "binning resize callback returned NULL"; return -3; }
        point_list = (uint32_t*)bin;
        entry_gauss = (uint32_t*)bin;
    }

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem;
    ba.n_tiles_local = (T - tile_rem + tile_mod - 1) / tile_mod;
    ba.ranges = ranges; ba.point_list = point_list; ba.rec = rec; ba.entry_gauss = entry_gauss;
    uint32_t* order = (uint32_t*)(img + IL.order);
    if (ba.n_tiles_local > 0) {
        ProfileScope ps(ST_RANGES, stream);
        uint32_t* ok = (uint32_t*)(img + IL.order_tmp_keys);
        uint32_t* ov = (uint32_t*)(img + IL.order_tmp_vals);
        hipLaunchKernelGGL(tile_order_keys_kernel, dim3((ba.n_tiles_local + 255) / 256), dim3(256), 0, stream, ba.n_tiles_local, tile_mod,
                           tile_rem, ranges, ok, ov);
        size_t tb3 = IL.sort_temp_bytes;
        GS_CHECK(rocprim::radix_sort_pairs(img + IL.sort_temp, tb3, ok, (uint32_t*)(img + IL.order_keys), ov, order,
                                           (size_t)ba.n_tiles_local, 0, 32, stream));
    }
    ba.order = order;
    ba.bg = background;
    ba.out_color = out_color; ba.out_depth = out_depth; ba.final_T = final_T; ba.n_contrib = n_contrib; ba.is_used = is_used;
    (void)HW;
    if (ba.n_tiles_local > 0) {
        ProfileScope ps(ST_BLEND_FWD, stream);
        hipLaunchKernelGGL(blend_forward_strip_kernel, dim3(ba.n_tiles_local * 4), dim3(64), 0, stream, ba);
    }
    GS_CHECK(hipGetLastError());
    return num_rendered;
}

size_t gsicp_raster_backward_scratch_bytes(int num_rendered, int width, int height) {
    const size_t T = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    const size_t R = num_rendered > 0 ? (size_t)num_rendered : 1;
    (void)T;
    return align_up(R * 4 * SLOT_F * sizeof(float)) + align_up(R * SLOT_F * sizeof(float));
}

int gsicp_raster_backward(int P, int D, int M, int num_rendered, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                          const char* geom_buffer, const char* binning_buffer, const char* img_buffer, char* scratch,
                          const float* dL_dpix, const float* dL_ddepth, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity,
                          float* dL_dcolors, float* dL_ddepths, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscales, float* dL_drots, int tile_mod, int tile_rem, int debug, void* stream_v) {
    (void)debug;
    hipStream_t stream = (hipStream_t)stream_v;
    if (P <= 0) return 0;
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "bad tile_mod / tile_rem"; return -2; }
    if (!scratch) { g_last_error = "gsicp_raster_backward: scratch buffer is NULL (size it with gsicp_raster_backward_scratch_bytes)"; return -2; }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, T = gx * gy;
    const GeomLayout GL = geom_layout(P, 0);
    const BinLayout BL = bin_layout((size_t)num_rendered, 0);
    const ImgLayout IL = img_layout(width, height);
    float* slots = (float*)scratch;
    float* entry_sum = (float*)(scratch + align_up((size_t)(num_rendered > 0 ? num_rendered : 1) * 4 * SLOT_F * sizeof(float)));

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem;
    ba.n_tiles_local = (T - tile_rem + tile_mod - 1) / tile_mod;
    ba.ranges = (const uint2*)(img_buffer + IL.ranges);
    ba.order = (const uint32_t*)(img_buffer + IL.order);
    ba.point_list = (const uint32_t*)(binning_buffer + BL.point_list);
    ba.entry_gauss = (const uint32_t*)(binning_buffer + BL.entry_gauss);
    ba.rec = (const SplatRec*)(geom_buffer + GL.records);
    ba.bg = background;
    ba.final_T = (float*)(img_buffer + IL.final_T);
    ba.n_contrib = (uint32_t*)(img_buffer + IL.n_contrib);
    ba.dL_dpix = dL_dpix; ba.dL_ddepth = dL_ddepth;
    ba.slots = slots;
    if (num_rendered > 0 && ba.n_tiles_local > 0) {
        ProfileScope ps(ST_BLEND_BWD, stream);
        hipLaunchKernelGGL(blend_backward_strip_kernel, dim3(ba.n_tiles_local * 4), dim3(64), 0, stream, ba);
    }
    if (num_rendered > 0) {
        ProfileScope ps(ST_MEMSET, stream);   // stage name kept for the profile table: "entry gradient sum"
        hipLaunchKernelGGL(entry_sum_kernel, dim3((num_rendered + 255) / 256), dim3(256), 0, stream, num_rendered,
                           (const uint32_t*)(binning_buffer + BL.point_list_unsorted), slots, entry_sum);
    }
    {
        ProfileScope ps(ST_MEMSET, stream);
        const uint32_t* offsets = (const uint32_t*)(geom_buffer + GL.offsets);
        if (num_rendered > 0) {
            hipLaunchKernelGGL(gaussian_grad_gather_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P,
                               (const uint32_t*)(geom_buffer + GL.ids_sorted), offsets, entry_sum, dL_dmeans2D, dL_dconic, dL_dopacity,
                               dL_dcolors, dL_ddepths);
        } else {
            GS_CHECK(hipMemsetAsync(dL_dmeans2D, 0, (size_t)P * 12, stream));
            GS_CHECK(hipMemsetAsync(dL_dconic, 0, (size_t)P * 16, stream));
            GS_CHECK(hipMemsetAsync(dL_dopacity, 0, (size_t)P * 4, stream));
            GS_CHECK(hipMemsetAsync(dL_dcolors, 0, (size_t)P * 12, stream));
            GS_CHECK(hipMemsetAsync(dL_ddepths, 0, (size_t)P * 4, stream));
        }
    }

    PreprocessBwdArgs pb;
    pb.P = P; pb.D = D; pb.M = M; pb.W = width; pb.H = height;
    pb.means3D = means3D; pb.shs = shs; pb.colors_precomp = colors_precomp; pb.scales = scales; pb.rotations = rotations;
    pb.cov3D_precomp = cov3D_precomp; pb.scale_modifier = scale_modifier; pb.view = viewmatrix; pb.proj = projmatrix;
    pb.campos = cam_pos; pb.tanfovx = tan_fovx; pb.tanfovy = tan_fovy; pb.radii = radii;
    pb.clamped = (const unsigned char*)(geom_buffer + GL.clamped);
    pb.dL_dmean2D = dL_dmeans2D; pb.dL_dconic = dL_dconic; pb.dL_dopacity = dL_dopacity; pb.dL_dcolors = dL_dcolors;
    pb.dL_ddepths = dL_ddepths;
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
