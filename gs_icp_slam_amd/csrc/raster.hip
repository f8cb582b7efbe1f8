// gfx950 rasteriser: tile binning (R2-R5), front-to-back blend (R6), back-to-front gradient replay (R7) and the
// C-ABI entry points declared in include/gsicp_hip.h.
//
// Replaces diff_gaussian_rasterization._C.rasterize_gaussians / rasterize_gaussians_backward / mark_visible as called
// from [REF gaussian_renderer/__init__.py:294-302] and [REF mp_Mapper.py:242].
//
// MI355X-first design notes (details and measurements: DESIGN.md):
//  * Binning is "emit, multi-split, sort locally" instead of a global radix sort of all (tile, depth) duplicates (the
//    classic implementation: ~25 small launches through a sort library, each a dependent kernel boundary of 5-10 us on
//    this chip).  preprocess hands each Gaussian a private run of "emission slots" (block scan + one atomic per
//    workgroup); emit_kernel fills them; a three-kernel multi-split groups them by tile through per-workgroup LDS
//    histograms over all tiles (no global atomics, no cursor); one single-workgroup kernel turns the tile counts into
//    list ranges and an LPT dispatch order; one workgroup per tile then bitonic-sorts its list in LDS by (depth bits,
//    Gaussian id).  The (depth, id) order is total, so the lists are exactly those of a stable (tile << 32 | depth)
//    sort of duplicates emitted in id order — the parity tests compare them bit-for-bit.
//  * Blend kernels: one wave64 per (tile, 8x8 block) (rounds 1-2: 16x4 strips; the code keeps the word "strip").  Binning tags every
//    list entry with 4 bits — which blocks the splat's alpha >= 1/255 footprint can reach: the footprint's bounding box at emit time,
//    re-tested against the footprint ellipse itself when the tile sort writes the list (refine_block_bits) —; a wave compacts its
//    64-entry batches by its bit and stages only the surviving 48-byte records in LDS — no workgroup barriers, and culled entries cost
//    1/64 of a vector instruction.  Work items are dispatched longest list first (LPT).
//  * Backward: every lane of a wave walks the same splat at the same step, so the 10 partial sums are reduced across the
//    64 lanes with 16 permlane-swap folds + 7 bank-masked DPP adds; the four strip waves of a tile meet once per 64-entry
//    batch and write ONE 48-byte record per (tile, entry); each Gaussian then sums its own contiguous run of records.
//    No float atomics anywhere: device-scope atomics resolve at the memory side on this 8-XCD part, and gradients stay
//    bit-reproducible.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/gsicp_hip.h"
#include "raster_common.hpp"

namespace gsicp {

thread_local std::string g_last_error;
static std::atomic<int>& tile_sort_lds_flag() {     // GSICP_TILE_SORT_LDS=1 / gsicp_raster_set_tile_sort_lds: the all-LDS sorting network of rounds 2-5
    static std::atomic<int> v([] { const char* e = getenv("GSICP_TILE_SORT_LDS"); return (e && e[0] == '1') ? 1 : 0; }());
    return v;
}
// the counter region (inside the caller's img scratch) of this thread's LAST forward call: gsicp_raster_last_zero_region (pre-zeroed forward, round 6)
static thread_local void* g_last_zero_ptr = nullptr;
static thread_local size_t g_last_zero_words = 0;

// ------------------------------------------------------------------------------------------------ profiler
namespace {
struct ProfRec { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_log;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t g_prof_open[ST_COUNT];
std::atomic<int> g_prof_enabled{0};
// stage -> kernel: preprocess = preprocess_kernel; tile_scan_lpt = tile_scan_lpt_kernel; emit = emit_kernel; split_hist / split_colscan /
// split_scatter = the three multi-split kernels; tile_sort = tile_sort_kernel;
// blend_forward = blend_forward_strip_kernel; blend_backward = blend_backward_tile_kernel; preprocess_backward = preprocess_backward_kernel;
// gicp_* = the tracker's call-level stages (several launches each); loss_pass1 / loss_pass2 = the two loss kernels (pass 1 includes the
// one-workgroup reduce); adam = adam_tensor_kernel (+ the one-thread step bump)
const char* const g_stage_names[ST_COUNT] = {"preprocess", "tile_scan_lpt", "emit", "split_hist", "split_colscan", "split_scatter",
                                             "tile_sort", "blend_forward", "blend_backward", "preprocess_backward", "gicp_knn_cov",
                                             "gicp_grid_build", "gicp_align", "gicp_exact_nn", "loss_pass1", "loss_pass2", "adam", "entry_run_sum"};
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

// The number of (Gaussian, tile) duplicates R lives in device memory (preprocess counts it).  Every binning kernel
// reads it there, so the forward can run without a host round trip when the caller provides the list capacity
// (gsicp_raster_forward_async).  R above the capacity degrades to "nothing rendered" — memory-safe, and flagged to the
// caller through num_rendered_dev.
__device__ inline int device_R(const uint32_t* __restrict__ total, uint32_t cap) {
    const uint32_t r = *total;
    return r > cap ? 0 : (int)r;
}

// Zero fill of the per-call counters and of is_used, as a kernel: memset / memcpy NODES inside a replayed hipGraph proved
// unreliable on this stack (the counter region came back holding a constant garbage value after a device synchronise),
// so the capturable path contains kernel nodes only.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = 0u;
}
__global__ void publish_count_kernel(const uint32_t* __restrict__ total, uint32_t* __restrict__ out) { *out = total ? *total : 0u; }

// ------------------------------------------------------------------------------------------------ binning
// Single workgroup: exclusive scan of the per-tile duplicate counts -> list ranges, and an LPT dispatch order by a
// counting sort over 64 length buckets (exact ordering is not needed for load balance).  Any T; one launch.
constexpr int LPT_BUCKETS = 64;   // = the wave size (the bucket scan below runs on one wave)
template <int THREADS>
__device__ __forceinline__ void tile_scan_lpt_body(int T, int gx, int tile_mod, int tile_rem, const uint32_t* tile_count, uint2* __restrict__ ranges,
                                                   uint32_t* __restrict__ order, uint32_t* s_wave /* [THREADS / 64] */, uint32_t* s_hist /* [LPT_BUCKETS] */,
                                                   uint32_t* s_maxlen_p) {
    uint32_t& s_maxlen = *s_maxlen_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_maxlen = 0;
    if (tid < LPT_BUCKETS) s_hist[tid] = 0;
    __syncthreads();
    // exclusive scan in ONE pass: thread i owns the `per` consecutive tiles starting at i * per (one wave scan + 16 wave totals, two barriers,
    // whatever T is; a 1024-tile-per-round loop cost three barriers per round)
    {
        const int per = (T + THREADS - 1) / THREADS;
        const int t0 = tid * per, t1 = (t0 + per) < T ? (t0 + per) : T;
        uint32_t sum = 0, wmax = 0;
        for (int t = t0; t < t1; ++t) { const uint32_t c = tile_count[t]; sum += c; wmax = c > wmax ? c : wmax; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(wmax, off, 64); wmax = o > wmax ? o : wmax; }
        if (lane == 63) s_wave[wave] = incl;
        if (lane == 0 && wmax > 0) atomicMax(&s_maxlen, wmax);
        __syncthreads();
        uint32_t start = incl - sum;
        for (int w = 0; w < wave; ++w) start += s_wave[w];
        for (int t = t0; t < t1; ++t) {
            const uint32_t c = tile_count[t];
            ranges[t] = c ? make_uint2(start, start + c) : make_uint2(0u, 0u);   // empty tiles read (0,0), as the reference leaves them
            start += c;
        }
    }
    // LPT order over this rank's tiles
    const uint32_t maxlen = s_maxlen;
    const uint32_t div = maxlen / LPT_BUCKETS + 1;
    for (int t = tid; t < T; t += THREADS) {
        if (!tile_is_mine(t, gx, tile_mod, tile_rem)) continue;
        const uint32_t c = tile_count[t];
        atomicAdd(&s_hist[LPT_BUCKETS - 1 - (c / div)], 1u);     // bucket 0 = longest lists
    }
    __syncthreads();
    if (tid < LPT_BUCKETS) {   // exclusive scan of the 64 bucket counts on the first wave (LPT_BUCKETS == 64)
        const uint32_t h = s_hist[tid];
        uint32_t incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        s_hist[tid] = incl - h;
    }
    __syncthreads();
    for (int t = tid; t < T; t += THREADS) {
        if (!tile_is_mine(t, gx, tile_mod, tile_rem)) continue;
        const uint32_t c = tile_count[t];
        const uint32_t pos = atomicAdd(&s_hist[LPT_BUCKETS - 1 - (c / div)], 1u);
        order[pos] = (uint32_t)t;
    }
}
__global__ __launch_bounds__(1024) void tile_scan_lpt_kernel(int T, int gx, int tile_mod, int tile_rem, const uint32_t* __restrict__ tile_count,
                                                             uint2* __restrict__ ranges, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_hist[LPT_BUCKETS];
    __shared__ uint32_t s_maxlen;
    tile_scan_lpt_body<1024>(T, gx, tile_mod, tile_rem, tile_count, ranges, order, s_wave, s_hist, &s_maxlen);
}

// One thread per Gaussian (id order): fill its contiguous run of emission slots — tile id, depth bits, list word
// (slot | strip bits), slot -> Gaussian map.  Coalesced-ish plain stores, no atomics.  A splat that covers more than EMIT_WIDE tiles (a
// near-field or badly conditioned Gaussian can cover the whole screen: thousands of tiles) is not walked by its own thread — that would
// stall its wave for the length of the longest splat — but handed to the whole wave afterwards, one tile per lane and trip.
constexpr int EMIT_WIDE = 48;
__device__ inline void emit_one(int x, int y, int gx, int tile_mod, int tile_rem, float fx0, float fx1, float fy0, float fy1, uint32_t dbits,
                                uint32_t id, uint32_t u, uint32_t* __restrict__ emit_tile, uint32_t* __restrict__ emit_depth,
                                uint32_t* __restrict__ entry_gauss, uint32_t* __restrict__ entry_bits) {
    const int t = y * gx + x;
    // bit b: the footprint box can touch the 8x8 pixel BLOCK b of the tile (b & 1 = right half, b >> 1 = lower half).  Square blocks are cut
    // by fewer footprints than 16x4 strips of the same area: 1.60 instead of 1.71 kept (block, entry) pairs per list entry on the benchmark
    // scene, 1.54 instead of 1.67 at 640x480 (round 3, counted on the CPU from the oracle's lists).
    uint32_t bits = 0;
    const float xl = (float)(x * TILE), yl = (float)(y * TILE);
    const bool x0_ = fx1 >= xl && fx0 <= xl + 7.f, x1_ = fx1 >= xl + 8.f && fx0 <= xl + 15.f;
    const bool y0_ = fy1 >= yl && fy0 <= yl + 7.f, y1_ = fy1 >= yl + 8.f && fy0 <= yl + 15.f;
    if (x0_ && y0_) bits |= 1u;
    if (x1_ && y0_) bits |= 2u;
    if (x0_ && y1_) bits |= 4u;
    if (x1_ && y1_) bits |= 8u;
    emit_tile[u] = (uint32_t)t;
    emit_depth[u] = dbits;
    entry_gauss[u] = id;
    entry_bits[u] = bits;
}
__global__ __launch_bounds__(256) void emit_kernel(int P, const uint32_t* __restrict__ total, uint32_t cap,
                                                   const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ slot_base,
                                                   const SplatRec* __restrict__ rec, const int* __restrict__ radii, int gx, int gy,
                                                   int tile_mod, int tile_rem, uint32_t* __restrict__ emit_tile,
                                                   uint32_t* __restrict__ emit_depth, uint32_t* __restrict__ entry_gauss,
                                                   uint32_t* __restrict__ entry_bits, uint32_t* __restrict__ num_rendered_dev) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (id == 0 && num_rendered_dev) *num_rendered_dev = *total;   // the caller's copy of R (async path)
    const bool active = id < P && tiles_touched[id] != 0 && *total <= cap;
    uint32_t u = 0, dbits = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    float fx0 = 0.f, fx1 = 0.f, fy0 = 0.f, fy1 = 0.f;
    if (active) {
        u = slot_base[id];
        const SplatRec r = rec[id];
        dbits = __float_as_uint(r.depth);
        tile_rect(r.px, r.py, radii[id], gx, gy, x0, y0, x1, y1);
        fx0 = r.px - r.hx; fx1 = r.px + r.hx; fy0 = r.py - r.hy; fy1 = r.py + r.hy;   // alpha footprint box
    }
    const int area = (x1 - x0) * (y1 - y0);
    const bool wide = active && area > EMIT_WIDE;
    if (active && !wide) {
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                if (!tile_xy_is_mine(x, y, gx, tile_mod, tile_rem)) continue;
                emit_one(x, y, gx, tile_mod, tile_rem, fx0, fx1, fy0, fy1, dbits, (uint32_t)id, u, emit_tile, emit_depth, entry_gauss, entry_bits);
                ++u;
            }
    }
    // wide splats: the wave takes them one at a time, a tile per lane (slots stay in row-major tile order, as the per-thread walk emits them)
    for (unsigned long long m = __ballot(wide); m; m &= m - 1) {
        const int src = __ffsll((long long)m) - 1;
        const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bx1 = __shfl(x1, src, 64), by1 = __shfl(y1, src, 64);
        const float gfx0 = __shfl(fx0, src, 64), gfx1 = __shfl(fx1, src, 64), gfy0 = __shfl(fy0, src, 64), gfy1 = __shfl(fy1, src, 64);
        const uint32_t gdb = (uint32_t)__shfl((int)dbits, src, 64), gu = (uint32_t)__shfl((int)u, src, 64);
        const uint32_t gid = (uint32_t)(blockIdx.x * 256 + (threadIdx.x & ~63) + src);
        const int w = bx1 - bx0, n = w * (by1 - by0);
        if (tile_mod == 1) {
            for (int k = lane; k < n; k += 64)
                emit_one(bx0 + k % w, by0 + k / w, gx, tile_mod, tile_rem, gfx0, gfx1, gfy0, gfy1, gdb, gid, gu + (uint32_t)k, emit_tile, emit_depth,
                         entry_gauss, entry_bits);
        } else {   // sharded: only this rank's tiles own slots; rank them with a running count of the tiles kept so far
            uint32_t kept = 0;
            for (int k0 = 0; k0 < n; k0 += 64) {
                const int k = k0 + lane;
                const bool mine = k < n && tile_xy_is_mine(bx0 + k % w, by0 + k / w, gx, tile_mod, tile_rem);
                const unsigned long long mm = __ballot(mine);
                if (mine)
                    emit_one(bx0 + k % w, by0 + k / w, gx, tile_mod, tile_rem, gfx0, gfx1, gfy0, gfy1, gdb, gid,
                             gu + kept + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull)), emit_tile, emit_depth, entry_gauss, entry_bits);
                kept += (uint32_t)__popcll(mm);
            }
        }
    }
}

// Round 5, unsharded images (tile_mod == 1): the same slots, filled SLOT-PARALLEL.  emit_kernel walks a Gaussian's run with one thread — at step j
// the 64 lanes of a wave store to 64 runs' j-th slots, ~30 B apart (a run is 7.5 slots on a trained map): ~30 cache lines per store instruction
// instead of 4, four arrays, as many steps as the wave's longest run (35), and on the S-map 82 % of the lanes hold a culled Gaussian.  Here a
// workgroup's 256 Gaussians publish their rectangle, footprint box and depth bits in LDS together with the inclusive scan of their run lengths;
// the runs of one workgroup are contiguous and in thread order (preprocess_kernel allocates them that way, and this kernel uses the same
// 256-Gaussian blocking), so thread t then fills slots t, t + 256, ... of the workgroup's range: binary search of the slot in the scan (8 LDS
// reads), tile = the (slot - run start)-th of the rectangle in row-major order — the order the per-thread walk emits.  Same words in the same
// slots; every store instruction writes 64 consecutive words.
__global__ __launch_bounds__(256) void emit_flat_kernel(int P, const uint32_t* __restrict__ total, uint32_t cap,
                                                        const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ slot_base,
                                                        const SplatRec* __restrict__ rec, const int* __restrict__ radii, int gx, int gy,
                                                        uint32_t* __restrict__ emit_tile, uint32_t* __restrict__ emit_depth, uint32_t* __restrict__ entry_gauss,
                                                        uint32_t* __restrict__ entry_bits, uint32_t* __restrict__ num_rendered_dev) {
    __shared__ uint32_t s_incl[256], s_cnt[256], s_dbits[256], s_wave_tot[4], s_base0;
    __shared__ int s_x0[256], s_y0[256], s_w[256];
    __shared__ float s_fx0[256], s_fx1[256], s_fy0[256], s_fy1[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int id = blockIdx.x * 256 + tid;
    if (id == 0 && num_rendered_dev) *num_rendered_dev = *total;   // the caller's copy of R (async path)
    const bool active = id < P && tiles_touched[id] != 0 && *total <= cap;
    uint32_t cnt = 0;
    if (tid == 0) s_base0 = 0u;
    if (active) {
        cnt = tiles_touched[id];
        const SplatRec r = rec[id];
        int x0, y0, x1, y1;
        tile_rect(r.px, r.py, radii[id], gx, gy, x0, y0, x1, y1);
        s_x0[tid] = x0; s_y0[tid] = y0; s_w[tid] = x1 - x0;
        s_dbits[tid] = __float_as_uint(r.depth);
        s_fx0[tid] = r.px - r.hx; s_fx1[tid] = r.px + r.hx; s_fy0[tid] = r.py - r.hy; s_fy1[tid] = r.py + r.hy;   // alpha footprint box
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave_tot[w];
    incl += wave_off;
    s_incl[tid] = incl;
    s_cnt[tid] = cnt;
    if (active) s_base0 = slot_base[id] - (incl - cnt);          // the same value from every active thread: the workgroup's first slot
    __syncthreads();
    const uint32_t tot = s_incl[255], base0 = s_base0;
    for (uint32_t k = (uint32_t)tid; k < tot; k += 256u) {
        int lo = 0;                                               // number of Gaussians whose runs end at or before slot k = the Gaussian that owns it
#pragma unroll
        for (int step = 128; step > 0; step >>= 1)
            if (s_incl[lo + step - 1] <= k) lo += step;
        const uint32_t off = k - (s_incl[lo] - s_cnt[lo]);
        const int w = s_w[lo];
        const int x = s_x0[lo] + (int)(off % (uint32_t)w), y = s_y0[lo] + (int)(off / (uint32_t)w);
        emit_one(x, y, gx, 1, 0, s_fx0[lo], s_fx1[lo], s_fy0[lo], s_fy1[lo], s_dbits[lo], (uint32_t)(blockIdx.x * 256 + lo), base0 + k, emit_tile,
                 emit_depth, entry_gauss, entry_bits);
    }
}

// Tile multi-split, pass 1: each workgroup histograms its contiguous chunk of emission slots over all T tiles in LDS
// (LDS atomics; no global atomics) and writes its row of the (split block, tile) count table.
__global__ __launch_bounds__(1024) void split_hist_kernel(const uint32_t* __restrict__ total, uint32_t cap, int T,
                                                          const uint32_t* __restrict__ emit_tile, uint32_t* __restrict__ block_hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist_dyn[];
    const int R = device_R(total, cap), chunk = (R + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int t = threadIdx.x; t < T; t += 1024) s_hist_dyn[t] = 0;
    __syncthreads();
    const int lo = blockIdx.x * chunk, hi = (lo + chunk) < R ? (lo + chunk) : R;
    for (int u = lo + threadIdx.x; u < hi; u += 1024) atomicAdd(&s_hist_dyn[emit_tile[u]], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 1024) block_hist[(size_t)blockIdx.x * T + t] = s_hist_dyn[t];
}
// pass 2: one thread per tile turns its column of the table into exclusive prefixes and yields the tile's total
__device__ __forceinline__ void split_colscan_column(int T, int nb, int t, uint32_t* __restrict__ block_hist, uint32_t* __restrict__ tile_count) {
    uint32_t run = 0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {   // eight independent (coalesced across threads) loads in flight per step
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = block_hist[(size_t)(b + k) * T + t];
#pragma unroll
        for (int k = 0; k < 8; ++k) { block_hist[(size_t)(b + k) * T + t] = run; run += c[k]; }
    }
    for (; b < nb; ++b) {
        const uint32_t c = block_hist[(size_t)b * T + t];
        block_hist[(size_t)b * T + t] = run;
        run += c;
    }
    tile_count[t] = run;
}
// Round 5 experiment (`done` non-NULL: GSICP_TILE_SCAN_MERGED=1, default OFF): the LAST workgroup of this launch to finish also turns the tile totals
// into list ranges and the LPT order (tile_scan_lpt_body), so that the single-workgroup tile_scan_lpt_kernel is no launch of its own.  `done` is a
// zeroed word of the forward's counter block; the ~13 workgroups count themselves in after an agent-scope fence (their tile totals must be visible
// to the last one).  LOST: on gfx950 such a fence writes back the XCD's L2 — 21.9 us for this kernel against 7.4 + 7.6 us for the two launches.
__global__ __launch_bounds__(256) void split_colscan_kernel(int T, int nb, uint32_t* __restrict__ block_hist, uint32_t* tile_count, uint32_t* done, int gx,
                                                            int tile_mod, int tile_rem, uint2* __restrict__ ranges, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_hist[LPT_BUCKETS];
    __shared__ uint32_t s_maxlen;
    __shared__ int s_last;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < T) split_colscan_column(T, nb, t, block_hist, tile_count);
    if (!done) return;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done, 1u) == gridDim.x - 1u ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    tile_scan_lpt_body<256>(T, gx, tile_mod, tile_rem, tile_count, ranges, order, s_wave, s_hist, &s_maxlen);
}
// pass 3: scatter every emission slot into its tile's range; the rank inside the (block, tile) cell comes from an LDS
// cursor.  Order inside a tile is arbitrary here — the per-tile sort that follows makes it unique.
__global__ __launch_bounds__(1024) void split_scatter_kernel(const uint32_t* __restrict__ total, uint32_t cap, int T, const uint32_t* __restrict__ emit_tile,
                                                             const uint32_t* __restrict__ emit_depth, const uint32_t* __restrict__ entry_bits,
                                                             const uint32_t* __restrict__ block_hist, const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ entry_gauss, uint4* __restrict__ sc_pack) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist_dyn[];
    const int R = device_R(total, cap), chunk = (R + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int t = threadIdx.x; t < T; t += 1024) s_hist_dyn[t] = 0;
    __syncthreads();
    const int lo = blockIdx.x * chunk, hi = (lo + chunk) < R ? (lo + chunk) : R;
    for (int u = lo + threadIdx.x; u < hi; u += 1024) {
        const uint32_t t = emit_tile[u];
        const uint32_t r = atomicAdd(&s_hist_dyn[t], 1u);
        const uint32_t pos = ranges[t].x + block_hist[(size_t)blockIdx.x * T + t] + r;
        // round 5: depth bits, list word AND the Gaussian id in one 16-byte store (rounds 1-4: two scattered 4-byte stores here, and a dependent
        // 4-byte gather of entry_gauss[slot] per entry in the per-tile sort)
        sc_pack[pos] = make_uint4(emit_depth[u], (uint32_t)u | (entry_bits[u] << STRIP_SHIFT), entry_gauss[u], 0u);
    }
}

// Bitonic network over `npad` (power of two) keys in LDS, one compare-exchange PAIR per thread and step: pair p exchanges elements
// i = 2 j (p / j) + (p % j) and i + j.  64 consecutive pairs — one wave's share of a step — cover 128 consecutive elements, so every step
// with j < 64 only touches elements of the wave's own 128-element blocks and needs no workgroup barrier (a wave's LDS operations execute in
// program order): of the log2(n) (log2(n) + 1) / 2 steps only those with j >= 64 synchronise the workgroup (none up to 128 elements, 10 of 55
// at 1024).
template <int THREADS>
__device__ inline void bitonic_pairs(unsigned long long* __restrict__ s_key, uint32_t* __restrict__ s_val, int npad, int tid) {
    const int half = npad >> 1;
    int lk = 1;                                   // log2(k)
    for (int k = 2; k <= npad; k <<= 1, ++lk) {
        int lj = lk - 1;                          // log2(j): j is a power of two, so p / j and p % j are a shift and a mask (round 6: the compiler cannot know
        for (int j = k >> 1; j > 0; j >>= 1, --lj) {   // that and emitted a 32-bit integer division per compare-exchange: 24.2 -> 21.1 us S-map, 62.5 -> 52.8 trained)
            for (int p = tid; p < half; p += THREADS) {
                const int i = ((p >> lj) << (lj + 1)) | (p & (j - 1)), l = i + j;
                const unsigned long long a = s_key[i], b = s_key[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s_key[i] = b; s_key[l] = a;
                    const uint32_t va = s_val[i]; s_val[i] = s_val[l]; s_val[l] = va;
                }
            }
            if (j >= 64 && THREADS > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        }
        // the next merge level starts with j = k: its first step reads across wave blocks whenever k >= 64
        if (k >= 64 && k < npad && THREADS > 64) __syncthreads();
    }
}

// One workgroup per tile: sort the tile's list by (depth bits, Gaussian id) — a total order, so the result is unique.
// Lists up to CAP (= SORT_SMALL) entries are bitonic-sorted in LDS; longer ones are sorted in CAP-entry chunks and merged by rank
// (binary searches across the sorted chunks) — not reached by the scenes in BASELINE.json, whose longest lists are a few hundred entries.
constexpr int SORT_SMALL = 1024;   // LDS capacity of the sort kernel: 256 threads (one wave does all the work of a <= 128-entry list; the others only meet
                                   // it at three barriers), 12 KB LDS -> many workgroups per CU.  With the pair-per-thread network a 1024-entry list costs
                                   // ~6 us; longer lists (none in the BASELINE scenes) take the chunk-sort + rank-merge path below
// Exact block bits (round 3).  emit_one sets a block's bit when the BOUNDING BOX of the splat's alpha >= 1/255 footprint touches the block;
// here, one thread per list entry, every set bit is re-tested against the footprint itself: the ellipse q(d) = ca dx^2 + 2 cb dx dy + cc dy^2
// <= 2 tau (tau = ln(255 opacity) + 0.01: the 1 % pad in alpha that the box uses, far above any fp32 / exp rounding) meets the block iff the
// minimum of q over the rectangle of the block's pixel centres is <= 2 tau — the centre lies inside, or the minimum sits on one of the four
// edges (a clamped 1-D parabola each).  Conservative by construction (a continuous rectangle contains its 64 pixel centres), so the per-pixel
// tests still decide everything; it only removes (block, entry) pairs no pixel would accept: 1.25 instead of 1.60 per list entry on the
// benchmark scene (the pixel-exact count is 1.250) — a fifth of the blend kernels' evaluations.
__device__ inline uint32_t refine_block_bits(const uint32_t word, const SplatRec& r, const float tile_x0, const float tile_y0) {
    uint32_t bits = word >> STRIP_SHIFT;
    if (bits == 0u) return word;
    const float two_tau = 2.f * (__logf(255.f * r.opacity) + 0.01f);
    const float inv_ca = __builtin_amdgcn_rcpf(r.ca), inv_cc = __builtin_amdgcn_rcpf(r.cc);
    uint32_t out = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (!((bits >> b) & 1u)) continue;
        const float x0 = tile_x0 + (float)((b & 1) * 8), y0 = tile_y0 + (float)((b >> 1) * 8);
        // d = splat centre - pixel; over the block dx in [dx_lo, dx_hi], dy in [dy_lo, dy_hi]
        const float dx_hi = r.px - x0, dx_lo = dx_hi - 7.f, dy_hi = r.py - y0, dy_lo = dy_hi - 7.f;
        bool hit = dx_lo <= 0.f && dx_hi >= 0.f && dy_lo <= 0.f && dy_hi >= 0.f;     // the centre is inside the block: q = 0 there
        if (!hit) {
            float qmin = 3.4e38f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {        // vertical edges dx = const: the parabola in dy has its vertex at -cb dx / cc
                const float dx = e ? dx_hi : dx_lo;
                float dy = -r.cb * dx * inv_cc;
                dy = fminf(fmaxf(dy, dy_lo), dy_hi);
                qmin = fminf(qmin, r.ca * dx * dx + 2.f * r.cb * dx * dy + r.cc * dy * dy);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {        // horizontal edges dy = const
                const float dy = e ? dy_hi : dy_lo;
                float dx = -r.cb * dy * inv_ca;
                dx = fminf(fmaxf(dx, dx_lo), dx_hi);
                qmin = fminf(qmin, r.ca * dx * dx + 2.f * r.cb * dx * dy + r.cc * dy * dy);
            }
            hit = qmin <= two_tau;
        }
        if (hit) out |= 1u << b;
    }
    return (word & ID_MASK) | (out << STRIP_SHIFT);
}


// ---- Round 6: the tile sort's wave-local steps IN REGISTERS ------------------------------------------------------------------------------------------
// A wave owns 128-element blocks of the padded list; lane l holds elements 64 h + l (h = 0, 1) of a block as (64-bit key, list word).  Every compare-exchange
// with partner distance j < 64 is then a lane exchange (quad permutes, bank-masked row rotates, gfx950's v_permlane16/32_swap — no LDS round trip, no barrier),
// j = 64 is a swap between a lane's own two elements, and only the steps with j >= 128 (none for lists up to 128 entries, 1 / 3 / 6 of the 36 / 45 / 55 steps at
// 256 / 512 / 1024) go through LDS.  Same network, same total order on (depth bits, Gaussian id): the lists are the same bits.
typedef unsigned ts_uint2 __attribute__((ext_vector_type(2)));
template <int J>
__device__ __forceinline__ unsigned ts_lane_xor(unsigned v) {      // value of lane (l ^ J)
    const int iv = (int)v;
    if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, iv, 0x12C, 0xF, 0xF, true);                            // row_ror 12: lane l <- (l + 4) mod 16, right for banks 0, 2
        return (unsigned)__builtin_amdgcn_update_dpp(t, iv, 0x124, 0xF, 0xA, false);                        // banks 1, 3: row_ror 4, lane l <- l - 4
    } else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, true);   // row_ror 8
    else if constexpr (J == 16) {
        const ts_uint2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);    // x: rows (0, 0, 2, 2) of v, y: rows (1, 1, 3, 3)
        return __builtin_amdgcn_inverse_ballot_w64(0xFFFF0000FFFF0000ull) ? r.x : r.y;
    } else {
        static_assert(J == 32, "ts_lane_xor: unsupported pattern");
        const ts_uint2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // x: lower half of v twice, y: upper half twice
        return __builtin_amdgcn_inverse_ballot_w64(0xFFFFFFFF00000000ull) ? r.x : r.y;
    }
}
// one compare-exchange of element (key, val) with the same element slot of lane (lane ^ J); `up`: this element's merge block sorts ascending
template <int J>
__device__ __forceinline__ void ts_cx_lane(unsigned long long& key, uint32_t& val, const bool up, const int lane) {
    const unsigned long long pk = ((unsigned long long)ts_lane_xor<J>((unsigned)(key >> 32)) << 32) | (unsigned long long)ts_lane_xor<J>((unsigned)key);
    const uint32_t pv = ts_lane_xor<J>(val);
    const bool keep_min = (((lane & J) == 0) == up);
    const bool take = keep_min ? (pk < key) : (pk > key);
    key = take ? pk : key;
    val = take ? pv : val;
}

template <int CAP, int THREADS, int MIN_N>
__device__ inline void tile_sort_one(unsigned long long* __restrict__ s_key, uint32_t* __restrict__ s_val, const uint32_t tile,
                                     const uint2* __restrict__ ranges, const uint4* __restrict__ sc_pack,
                                     uint32_t* __restrict__ point_list,
                                     uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ list_gauss, const SplatRec* __restrict__ rec, const int gx,
                                     const int lds_only) {
    const uint2 range = ranges[tile];
    const float tile_x0 = (float)((int)(tile % (uint32_t)gx) * TILE), tile_y0 = (float)((int)(tile / (uint32_t)gx) * TILE);
    const int n = (int)(range.y - range.x);
    if (n <= MIN_N) return;   // (MIN_N > 0: a size class that leaves the short lists to another launch; not used by the shipped path)
    const int tid = threadIdx.x;
    if (n > CAP) {
        // Long list (> CAP entries: far beyond the BASELINE scenes, whose longest lists are a few hundred entries).  Chunks of CAP entries
        // are bitonic-sorted in LDS and written back in place; because (depth bits, Gaussian id) is a TOTAL order with unique keys, an
        // entry's final rank is its rank inside its own chunk plus, for every other chunk, the number of that chunk's keys below it —
        // one binary search per other chunk.  O(n (n / CAP) log CAP) instead of the O(n^2) rank sort this replaces, and still exact.
        uint4* gp = const_cast<uint4*>(sc_pack) + range.x;        // forward-only scratch: reordered in place
        const int n_chunks = (n + CAP - 1) / CAP;
        for (int c = 0; c < n_chunks; ++c) {
            const int base = c * CAP, m = (n - base) < CAP ? (n - base) : CAP;
            for (int i = tid; i < CAP; i += THREADS) {
                if (i < m) {
                    const uint4 e = gp[base + i];
                    s_val[i] = e.y;
                    s_key[i] = ((unsigned long long)e.x << 32) | e.z;
                } else {
                    s_key[i] = ~0ull;
                    s_val[i] = 0;
                }
            }
            __syncthreads();
            bitonic_pairs<THREADS>(s_key, s_val, CAP, tid);
            __syncthreads();
            for (int i = tid; i < m; i += THREADS) gp[base + i] = make_uint4((uint32_t)(s_key[i] >> 32), s_val[i], (uint32_t)s_key[i], 0u);
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < n; i += THREADS) {
            const uint4 e = gp[i];
            const uint32_t v = e.y, gid = e.z;
            const unsigned long long key = ((unsigned long long)e.x << 32) | gid;
            const int own = i / CAP;
            int rank = i - own * CAP;
            for (int c = 0; c < n_chunks; ++c) {
                if (c == own) continue;
                const int base = c * CAP, m = (n - base) < CAP ? (n - base) : CAP;
                int lo = 0, hi = m;                       // number of keys of chunk c below `key`
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const uint4 em = gp[base + mid];
                    const unsigned long long km = ((unsigned long long)em.x << 32) | em.z;
                    if (km < key) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            point_list[range.x + rank] = refine_block_bits(v, rec[gid], tile_x0, tile_y0);
            tile_keys[range.x + rank] = tile;
            list_gauss[range.x + rank] = gid;
        }
        return;
    }
    int npad = 64;
    while (npad < n) npad <<= 1;
    if (THREADS == 256 && CAP == 1024 && !lds_only) {
        // ---- register path (round 6): blocks of 128 elements per wave, block b = wave + 4 r (r = 0, 1); element e = 128 b + 64 h + lane
        const int lane = tid & 63, wave = tid >> 6;
        unsigned long long key[2][2];
        uint32_t val[2][2];
        bool act[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            act[r] = 128 * (wave + 4 * r) < npad;                 // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 128 * (wave + 4 * r) + 64 * h + lane;
                key[r][h] = ~0ull; val[r][h] = 0u;
                if (act[r] && e < n) {
                    const uint4 w = sc_pack[range.x + e];
                    val[r][h] = w.y;
                    key[r][h] = ((unsigned long long)w.x << 32) | w.z;
                }
            }
        }
        // compare-exchange steps with partner distance j = jmax, jmax / 2, .. 1 (jmax <= 64) at merge size k, on every active element
        auto reg_steps = [&](const int k, const int jmax) {
            if (jmax >= 64) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (!act[r]) continue;
                    const bool up = ((128 * (wave + 4 * r)) & k) == 0;    // (k >= 128 here: the direction is the block's)
                    const bool sw = (key[r][0] > key[r][1]) == up;
                    const unsigned long long k0 = key[r][0]; const uint32_t v0 = val[r][0];
                    key[r][0] = sw ? key[r][1] : k0; val[r][0] = sw ? val[r][1] : v0;
                    key[r][1] = sw ? k0 : key[r][1]; val[r][1] = sw ? v0 : val[r][1];
                }
            }
#define GSICP_TS_STEP(J)                                                                                              \
            if (jmax >= J) {                                                                                          \
                _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                       \
                    if (!act[r]) continue;                                                                            \
                    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                   \
                        const int e = 128 * (wave + 4 * r) + 64 * h + lane;                                           \
                        ts_cx_lane<J>(key[r][h], val[r][h], (e & k) == 0, lane);                                      \
                    }                                                                                                 \
                }                                                                                                     \
            }
            GSICP_TS_STEP(32) GSICP_TS_STEP(16) GSICP_TS_STEP(8) GSICP_TS_STEP(4) GSICP_TS_STEP(2) GSICP_TS_STEP(1)
#undef GSICP_TS_STEP
        };
        const int kreg = npad < 128 ? npad : 128;
        for (int k = 2; k <= kreg; k <<= 1) reg_steps(k, k >> 1);            // every 128-block sorted (alternating directions) without touching LDS
        for (int k = 256; k <= npad; k <<= 1) {                              // merges across blocks: the steps with j >= 128 through LDS, the rest in registers again
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (!act[r]) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 128 * (wave + 4 * r) + 64 * h + lane;
                    s_key[e] = key[r][h]; s_val[e] = val[r][h];
                }
            }
            __syncthreads();
            const int half = npad >> 1;
            int lj = 31 - __builtin_clz((unsigned)(k >> 1));
            for (int j = k >> 1; j >= 128; j >>= 1, --lj) {
                for (int p = tid; p < half; p += THREADS) {
                    const int i = ((p >> lj) << (lj + 1)) | (p & (j - 1)), l = i + j;
                    const unsigned long long a = s_key[i], b = s_key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s_key[i] = b; s_key[l] = a;
                        const uint32_t va = s_val[i]; s_val[i] = s_val[l]; s_val[l] = va;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (!act[r]) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 128 * (wave + 4 * r) + 64 * h + lane;
                    key[r][h] = s_key[e]; val[r][h] = s_val[e];
                }
            }
            __syncthreads();                                                 // the next merge level (or the next tile) stores into the arrays again
            reg_steps(k, 64);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (!act[r]) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 128 * (wave + 4 * r) + 64 * h + lane;
                if (e < n) {
                    const uint32_t gid = (uint32_t)key[r][h];               // low word of the sort key = Gaussian id
                    point_list[range.x + e] = refine_block_bits(val[r][h], rec[gid], tile_x0, tile_y0);
                    tile_keys[range.x + e] = tile;
                    list_gauss[range.x + e] = gid;
                }
            }
        }
        return;
    }
    for (int i = tid; i < npad; i += THREADS) {
        if (i < n) {
            const uint4 e = sc_pack[range.x + i];
            s_val[i] = e.y;
            s_key[i] = ((unsigned long long)e.x << 32) | e.z;
        } else {
            s_key[i] = ~0ull;
            s_val[i] = 0;
        }
    }
    // a single wave needs no s_barrier: its LDS operations execute in program order
    if (THREADS == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    bitonic_pairs<THREADS>(s_key, s_val, npad, tid);
    __syncthreads();
    for (int i = tid; i < n; i += THREADS) {
        const uint32_t gid = (uint32_t)s_key[i];        // low word of the sort key = Gaussian id
        point_list[range.x + i] = refine_block_bits(s_val[i], rec[gid], tile_x0, tile_y0);
        tile_keys[range.x + i] = tile;
        list_gauss[range.x + i] = gid;
    }
}
// One workgroup per tile of the LPT order (grid-stride, so a smaller persistent grid also works; `limit_dev`, optional, bounds the
// number of tiles read from the device).
template <int CAP, int THREADS, int MIN_N>
__global__ __launch_bounds__(THREADS) void tile_sort_kernel(int n_tiles, const uint32_t* __restrict__ limit_dev, const uint32_t* __restrict__ order,
                                                            const uint2* __restrict__ ranges, const uint4* __restrict__ sc_pack,
                                                            uint32_t* __restrict__ point_list, uint32_t* __restrict__ tile_keys,
                                                            uint32_t* __restrict__ list_gauss, const SplatRec* __restrict__ rec, int gx,
                                                            int lds_only /* A/B + test hook: every step through LDS, as rounds 2-5 */) {
    __shared__ unsigned long long s_key[CAP];
    __shared__ uint32_t s_val[CAP];
    int limit = n_tiles;
    if (limit_dev) { const int l = (int)*limit_dev; limit = l < limit ? l : limit; }
    for (int bi = blockIdx.x; bi < limit; bi += gridDim.x) {
        tile_sort_one<CAP, THREADS, MIN_N>(s_key, s_val, order[bi], ranges, sc_pack, point_list, tile_keys, list_gauss, rec, gx, lds_only);
        __syncthreads();   // the LDS arrays are reused by the next list
    }
}

// ------------------------------------------------------------------------------------------------ blending
struct BlendArgs {
    int W, H, gx, n_tiles_local, tile_mod, tile_rem;
    int depth_mode;            // 0: D = sum z alpha T;  1: alpha-normalised, D = sum z alpha T / (1 - T_final)   (include/gsicp_hip.h)
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
    const uint32_t* list_gauss;    // Gaussian id of every list entry, in list order (a coalesced read next to point_list)
    // backward only
    const float* dL_dpix;
    const float* dL_ddepth;
    const float* depth_out;    // forward depth image (depth_mode 1 needs it: d(N/A) involves N/A)
    float* entry_sum;    // (R, SLOT_F) per-emission-slot gradient moment sums (see blend_backward_tile_kernel)
};

// ================================================================================================================
// "Strip" blend kernels (default; the word is kept from rounds 1-2, when a wave owned a 16x4 strip of rows).  A 16x16 tile is cut into four
// 8x8 pixel BLOCKS; one wave64 owns block s (s & 1: right half, s >> 1: lower half; lane = 8 * row + column inside the block) and runs on
// its own — in the forward as a 64-thread workgroup per (tile, block), no workgroup barrier anywhere.  Per 64 list entries a wave
//   1. loads the entries with one coalesced vector load and keeps those whose block bit (computed at binning time from the
//      Gaussian's alpha >= 1/255 footprint, then made exact against the block rectangle by refine_block_bits) is set — on the
//      benchmark map 1.6 of a tile's four blocks per entry;
//   2. compacts the survivors with ballot/mbcnt, gathers their 48-byte records with all lanes in parallel into a
//      wave-private LDS slab (the gather latency is paid once per batch, not once per entry);
//   3. walks the compacted slab with wave-uniform LDS broadcast reads.
// Entries that cannot touch the block therefore cost 1/64 of a vector instruction instead of a full evaluation, and
// the four waves never wait for each other.  Exactness is untouched: the per-pixel alpha / transmittance tests are
// still what decides, the block bit only removes entries every pixel of the block would have rejected anyway.
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
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);   // wave = 8x8 block of the tile
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
        uint32_t e = 0, id = 0;
        if (k < range.y) { e = a.point_list[k]; id = a.list_gauss[k]; }
        const bool keep = (e & strip_bit) != 0;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) continue;
        const int n = __popcll(m);
        if (keep) {
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            s_id[slot] = id;
            s_pos[slot] = k - range.x + 1;   // 1-based position in the tile list (the reference's "contributor")
            s_rec[slot] = a.rec[id];
        }
        __builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in order; this only stops compiler reordering
        unsigned long long used = 0ull;    // wave-uniform bookkeeping in scalar registers: bit j = compacted entry j reached a pixel
        // Branch-free per-entry body: an entry this pixel does not blend enters the sums with weight 0 and leaves T where it was;
        // only the wave-uniform "nobody blends it" case branches.  Same arithmetic as the reference order of tests: alpha, then the
        // transmittance test on T (1 - alpha), then the accumulation with w = alpha T.
        auto eval = [&](const SplatRec& r, const uint32_t pos_v, const int j) {
            const float dx = r.px - pfx, dy = r.py - pfy;
            const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
            const float alpha = fminf(0.99f, r.opacity * __expf(power));
            const bool valid = !done && power <= 0.f && alpha >= 1.f / 255.f;
            const float test_T = T * (1.f - alpha);
            const bool stop = valid && test_T < 0.0001f;
            done = done || stop;
            const bool contrib = valid && !stop;
            if (__ballot(contrib) == 0ull) return;
            const float w = contrib ? alpha * T : 0.f;
            // (FMAs: this file is built with the compiler's default contraction; written out so that the accumulation order is explicit)
            C0 = __builtin_fmaf(r.r, w, C0); C1 = __builtin_fmaf(r.g, w, C1); C2 = __builtin_fmaf(r.b, w, C2); Dz = __builtin_fmaf(r.depth, w, Dz);
            T = contrib ? test_T : T;
            last_contributor = contrib ? pos_v : last_contributor;
            used |= 1ull << j;
        };
        // two entries per trip with two named register sets (no "next -> current" copies; one set's LDS reads fly while the other is evaluated)
        SplatRec ra = s_rec[0];
        uint32_t pa = s_pos[0];
        for (int j = 0; j < n; j += 2) {
            const int j1 = j + 1 < n ? j + 1 : j;
            const SplatRec rb = s_rec[j1];
            const uint32_t pb = s_pos[j1];
            eval(ra, pa, j);
            const int j2 = j + 2 < n ? j + 2 : j1;
            ra = s_rec[j2];
            pa = s_pos[j2];
            if (j + 1 < n) eval(rb, pb, j + 1);
        }
        // is_used: one store instruction per batch (lane j reports compacted entry j) instead of a store per visit
        if (a.is_used && lane < n && ((used >> lane) & 1ull)) a.is_used[s_id[lane]] = 1;
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
        a.out_depth[pix] = a.depth_mode == 1 ? (T < 1.f ? Dz / (1.f - T) : 0.f) : Dz;
    }
}

// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows of TWO registers in one instruction, so two
// values can be folded into one register per level ("A keeps its lower half and receives B's lower half; B keeps its upper halves").
typedef unsigned gs_uint2 __attribute__((ext_vector_type(2)));
__device__ inline float swap32_add(float a, float b) {
    const gs_uint2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ inline float swap16_add(float a, float b) {
    const gs_uint2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// ================================================================================================================
// Backward, tile workgroups (default).  One 256-thread workgroup per tile; wave s owns the 8x8 block s exactly as in the forward (own
// compaction by block bit, own slab, own pixel state), but the four waves meet once per 64-entry batch so that every (tile, entry)
// gradient record is written ONCE:
//   * each wave reduces its ten partial gradients per staged entry across the 64 lanes and parks them in LDS OVER the entry's own,
//     already consumed Gaussian record (s_slot[wave][compacted slot]); two 64-bit masks per wave say which batch positions it kept / wrote;
//   * after one workgroup barrier, 192 threads add the (up to four) block records of each batch position in block order and store the
//     48-byte entry record straight into entry_sum[emission slot] — the array the per-Gaussian pass (preprocess_backward) streams.
// Against the per-(entry, block) slot scheme of round 1 this removes the slot buffer (240 B per duplicate), the entry_sum kernel that re-read it,
// and ~60 % of the gradient write traffic; results stay bit-reproducible (fixed reduction tree, fixed block order, no atomics).
// The per-entry body is branch-free: an entry a pixel does not blend enters with alpha = 0, which makes T, the behind-colour A and all
// ten gradient terms no-ops by arithmetic (T * rcp(1) = T, 0 * c + 1 * A = A, w = 0) instead of by exec masking + zero fills, and the
// behind-colour recurrence is applied eagerly (A <- alpha c + (1 - alpha) A after the entry) rather than lazily before the next one.
// ================================================================================================================
// Ten lane-partials -> ten wave totals in ONE register: two permlane-swap levels fold ten registers into three (10 -> 5 across the
// 32-lane halves, 5 (+ a zero) -> 3 across the row pairs; row k of q0 then holds value {0,2,1,3}[k], of q1 {4,6,5,7}[k], of q2 {8,-,9,-}[k]);
// the three registers
// are folded across the 16 lanes of a row with bank-masked DPP adds (two registers into one per level: banks 2,3 take one source's
// shr:8 fold, banks 0,1 the other's shl:8 fold), so that the quad reduction runs on a single register: 7 DPP adds instead of 12.
// Result: lanes with (lane & 3) == 0 of bank b in row k hold  b=2: value {0,2,1,3}[k];  b=0: value {4,6,5,7}[k];  b=3: half of value {8,8,9,9}[k].
__device__ inline float wave_sum10_banked(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8, float v9) {
    const float p0 = swap32_add(v0, v1), p1 = swap32_add(v2, v3), p2 = swap32_add(v4, v5), p3 = swap32_add(v6, v7), p4 = swap32_add(v8, v9);
    // the fifth pair is NOT folded across row pairs (that would cost a zero register, a swap and an add per entry): rows 0, 1 of p4 hold
    // value 8 and rows 2, 3 value 9; their two row totals land in record words 8 / 10 and 9 / 11, and the per-batch merge adds them
    float q0 = swap16_add(p0, p1), q1 = swap16_add(p2, p3), q2 = p4;
    float r, r2, t;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(r), "=&v"(r2), "=&v"(t)
        : "v"(q0), "v"(q1), "v"(q2));
    return t;
}

__global__ __launch_bounds__(256) void blend_backward_tile_kernel(BlendArgs a) {
    const int tl = (int)blockIdx.x;            // tiles are dispatched longest list first (LPT), see the forward
    if (tl >= a.n_tiles_local) return;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int tile = (int)a.order[tl];
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7), py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);   // wave = 8x8 block of the tile
    const bool inside = px < a.W && py < a.H;
    const float pfx = (float)px, pfy = (float)py;
    const uint2 range = a.ranges[tile];
    const uint32_t strip_bit = 1u << (STRIP_SHIFT + wave);
    const size_t HW = (size_t)a.W * a.H;
    const size_t pix = (size_t)py * a.W + px;

    // A compacted entry's 48-byte Gaussian record and the 48-byte partial-gradient record the wave produces from it share ONE LDS slot: the
    // record is in registers (read one pair ahead) before its entry is evaluated, and nothing reads it again.  12 KB less LDS per workgroup.
    union BwdSlot { SplatRec rec; float part[SLOT_F]; };
    static_assert(sizeof(SplatRec) == SLOT_F * sizeof(float), "record and partial record must alias exactly");
    __shared__ BwdSlot s_slot[4][SLAB];
    __shared__ int s_pos[4][SLAB];
    __shared__ uint32_t s_e[SLAB];
    __shared__ unsigned long long s_wrote[4], s_keep[4];

    const float T_final = inside ? a.final_T[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)a.n_contrib[pix] : 0;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dpd = 0.f;
    if (inside) {
        dp0 = a.dL_dpix[pix]; dp1 = a.dL_dpix[HW + pix]; dp2 = a.dL_dpix[2 * HW + pix];
        dpd = a.dL_ddepth ? a.dL_ddepth[pix] : 0.f;
    }
    float bg_dot = a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2;
    if (a.depth_mode == 1) {
        // D = N / A with A = 1 - T_final:  dD/dalpha_i = (dN/dalpha_i) / A - (N / A^2) T_final / (1 - alpha_i).  The first term is the
        // un-normalised rule fed with dL/dD / A; the second has exactly the form of the background term (-T_final / (1 - alpha_i) * x).
        const float A = 1.f - T_final;
        if (inside && A > 0.f) { bg_dot += dpd * a.depth_out[pix] / A; dpd = dpd / A; } else dpd = 0.f;
    }
    const float bgT = -T_final * bg_dot;
    float A0 = 0.f, A1 = 0.f, A2 = 0.f, Ad = 0.f;     // colour / depth composited BEHIND the current entry

    // where this lane's value of wave_sum10_banked belongs in a 12-float record (lanes with (lane & 3) == 0 of banks 0, 2, 3 store)
    const int srow = lane >> 4, sbank = (lane >> 2) & 3;
    const int perm = srow == 0 ? 0 : (srow == 1 ? 2 : (srow == 2 ? 1 : 3));
    const int st_idx = sbank == 2 ? perm : (sbank == 0 ? 4 + perm : 8 + perm);      // 8 + {0,2,1,3}: rows 0, 1 -> words 8, 10 (value 8), rows 2, 3 -> 9, 11 (value 9)
    const bool st_lane = (lane & 3) == 0 && sbank != 1;

    // entries at positions >= the strip's largest n_contrib reach no pixel of this strip
    int top0 = last_contributor;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(top0, off, 64);
        top0 = o > top0 ? o : top0;
    }
    const int total = (int)(range.y - range.x);
    for (int top = total; top > 0; top -= 64) {   // entries [top-64, top) back-to-front; lane 0 holds the last one
        const int posn = top - 1 - lane;          // 0-based list position of this lane's entry
        uint32_t e = 0, gid = 0;
        if (posn >= 0) { e = a.point_list[range.x + (uint32_t)posn]; gid = a.list_gauss[range.x + (uint32_t)posn]; }
        if (wave == 0) s_e[lane] = e;
        const bool keep = (e & strip_bit) != 0;
        const unsigned long long m = __ballot(keep);
        unsigned long long wrote = 0ull;
        if (m != 0ull && top - 64 < top0) {
            const int n = __popcll(m);
            if (keep) {
                const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // kept lanes below this one
                s_pos[wave][slot] = posn;
                s_slot[wave][slot].rec = a.rec[gid];
            }
            __builtin_amdgcn_wave_barrier();
            auto eval = [&](const SplatRec& r, const int position_v, const int j) {
                const int position = __builtin_amdgcn_readfirstlane(position_v);   // wave-uniform (LDS broadcast read): keep it scalar
                const float dx = r.px - pfx, dy = r.py - pfy;
                const float power = -0.5f * (r.ca * dx * dx + r.cc * dy * dy) - r.cb * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, r.opacity * G);
                const bool valid = position < last_contributor && power <= 0.f && alpha >= 1.f / 255.f;
                if (__ballot(valid) == 0ull) return;                       // wave-uniform: no pixel of the strip blends this entry
                const float av = valid ? alpha : 0.f;
                const float inv_one_m = __builtin_amdgcn_rcpf(1.f - av);   // alpha <= 0.99: one v_rcp_f32 (1 ulp); rcp(1) = 1 exactly
                T = T * inv_one_m;
                const float w = av * T;
                // gradient algebra with the FMAs written out (this file is built with the compiler's default contraction anyway)
                const float dot = __builtin_fmaf(r.depth - Ad, dpd, __builtin_fmaf(r.b - A2, dp2, __builtin_fmaf(r.g - A1, dp1, (r.r - A0) * dp0)));
                float dL_dalpha = __builtin_fmaf(dot, T, bgT * inv_one_m);
                dL_dalpha = valid ? dL_dalpha : 0.f;
                const float one_m = 1.f - av;
                A0 = __builtin_fmaf(av, r.r, one_m * A0); A1 = __builtin_fmaf(av, r.g, one_m * A1); A2 = __builtin_fmaf(av, r.b, one_m * A2);
                Ad = __builtin_fmaf(av, r.depth, one_m * Ad);
                // Everything that is constant per entry (conic, opacity, the pixel->NDC factors) is applied ONCE per Gaussian by
                // preprocess_backward; the lanes only form h = G dL/dalpha and its first and second moments in (dx, dy):
                //   dL/dopacity = S[h];  dL/dconic = -op (S[h dx dx] / 2, S[h dx dy], S[h dy dy] / 2);
                //   dL/dmean2D = op (-ca S[h dx] - cb S[h dy], -cc S[h dy] - cb S[h dx]) * (W/2, H/2)
                const float h = G * dL_dalpha;
                const float hx = dx * h, hy = dy * h;
                const float tsum = wave_sum10_banked(hx, hy, dx * hx, dx * hy, dy * hy, h, w * dp0, w * dp1, w * dp2, w * dpd);
                const int bp = top - 1 - position;                         // position inside this batch = the lane that loaded the entry
                if (st_lane) s_slot[wave][j].part[st_idx] = tsum;          // over the entry's own (already consumed) Gaussian record
                wrote |= 1ull << bp;
            };
            // two entries per trip with two named register sets: the LDS reads of one are in flight while the other is evaluated,
            // and no record has to be copied from a "next" to a "current" register set
            SplatRec ra = s_slot[wave][0].rec;
            int pa = s_pos[wave][0];
            for (int j = 0; j < n; j += 2) {
                const int j1 = j + 1 < n ? j + 1 : j;
                const SplatRec rb = s_slot[wave][j1].rec;                  // read BEFORE eval(ra) overwrites slot j (j1 == j on an odd tail)
                const int pb = j + 1 < n ? s_pos[wave][j1] : 0x7fffffff;   // odd tail: a position no pixel can blend (nothing is written)
                eval(ra, pa, j);
                const int j2 = j + 2 < n ? j + 2 : j1;
                ra = s_slot[wave][j2].rec;                                 // j2 > j + 1 or the loop ends: never a slot that holds partials and is used
                pa = s_pos[wave][j2];
                eval(rb, pb, j + 1);
            }
        }
        if (lane == 0) { s_wrote[wave] = wrote; s_keep[wave] = m; }
        __syncthreads();
        {   // merge the strips of every batch position in strip order and write the entry record once
            const int p = (int)(threadIdx.x >> 2), q = (int)(threadIdx.x & 3);
            if (top - 1 - p >= 0 && q < 3) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if ((s_wrote[w] >> p) & 1ull) {
                        const int slot = __popcll(s_keep[w] & ((1ull << p) - 1ull));   // compaction keeps the order: lane p's entry sits behind the kept lanes below it
                        const float4 v = *(const float4*)&s_slot[w][slot].part[4 * q];
                        if (q == 2) { acc.x += v.x + v.z; acc.y += v.y + v.w; }      // words 10, 11 = the second row halves of values 8, 9
                        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
                    }
                }
                *(float4*)(a.entry_sum + (size_t)(s_e[p] & ID_MASK) * SLOT_F + 4 * q) = acc;
            }
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace gsicp

using namespace gsicp;

extern "C" {

int gsicp_abi_version(void) { return GSICP_ABI_VERSION; }
int gsicp_raster_set_tile_sort_lds(int lds_only) { return tile_sort_lds_flag().exchange(lds_only ? 1 : 0); }
int gsicp_raster_last_zero_region(void** ptr, size_t* words) {
    if (!ptr || !words) return -2;
    *ptr = g_last_zero_ptr; *words = g_last_zero_words;
    return g_last_zero_ptr ? 0 : -1;
}
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
    const size_t T = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    const GeomLayout G = geom_layout(P);
    const BinLayout B = bin_layout((size_t)num_rendered, T);
    const ImgLayout I = img_layout(width, height);
    out[0] = G.total; out[1] = B.total; out[2] = I.total; out[3] = G.records; out[4] = B.point_list; out[5] = B.tile_keys;
    out[6] = I.ranges; out[7] = I.final_T; out[8] = I.n_contrib; out[9] = G.clamped; out[10] = B.entry_gauss; out[11] = B.entry_bits;
    return 0;
}

static int raster_forward_impl(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                         gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                         int height, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                         int* is_used, int tile_mod, int tile_rem, int debug, int depth_mode, int capacity, unsigned int* num_rendered_dev,
                         const int* live_rows, int raw_params, void* stream_v) {
    (void)prefiltered;
    if (depth_mode < 0 || depth_mode > 1) { g_last_error = "depth_mode: 0 = sum z alpha T, 1 = alpha-normalised"; return -2; }
    const bool async = capacity > 0 && P > 0;   // capacity given: no host round trip, R stays on the device
    // PRE-ZEROED forward (round 6; `debug` bit 1, sync-free path only): the caller guarantees that this call's counter region — the one the previous call with the
    // same buffers reported through gsicp_raster_last_zero_region — has been cleared in stream order ahead of this call (gsicp_mapper_select_view_zero: the
    // keyframe-selection launch of a captured mapper iteration does it); no zero-fill launch here, and the preprocess kernel clears is_used.
    const bool prezeroed = async && (debug & 2) != 0;
    if (capacity > (int)ID_MASK) { g_last_error = "capacity above 2^28 duplicates is not supported"; return -2; }
    hipStream_t stream = (hipStream_t)stream_v;
    if (width <= 0 || height <= 0 || P < 0) { g_last_error = "gsicp_raster_forward: bad sizes"; return -2; }
    if (P > (int)ID_MASK) { g_last_error = "more than 2^28 Gaussians are not supported"; return -2; }
    if (P > 0 && (shs == nullptr) == (colors_precomp == nullptr)) { g_last_error = "provide exactly one of shs / colors_precomp"; return -2; }
    if (P > 0 && ((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) {
        g_last_error = "provide exactly one of (scales, rotations) / cov3D_precomp"; return -2;
    }
    if (shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1))) { g_last_error = "sh degree / coefficient count mismatch"; return -2; }
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "bad tile_mod / tile_rem"; return -2; }
    if (!geom_alloc || !binning_alloc || !img_alloc) { g_last_error = "null resize callback"; return -2; }

    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, T = gx * gy;

    const ImgLayout IL = img_layout(width, height);
    char* img = img_alloc(img_user, IL.total);
    if (!img) { g_last_error = "img resize callback returned NULL"; return -3; }
    uint2* ranges = (uint2*)(img + IL.ranges);
    float* final_T = (float*)(img + IL.final_T);
    uint32_t* n_contrib = (uint32_t*)(img + IL.n_contrib);
    uint32_t* order = (uint32_t*)(img + IL.order);
    uint32_t* tile_count = (uint32_t*)(img + IL.tile_count);   // [T] counts, [T] cursors, [64] counters: one memset
    uint32_t* total_counter = tile_count + 2 * T;

    const GeomLayout GL = geom_layout(P);
    char* geom = geom_alloc(geom_user, GL.total);
    if (!geom) { g_last_error = "geom resize callback returned NULL"; return -3; }
    SplatRec* rec = (SplatRec*)(geom + GL.records);
    uint32_t* tiles_touched = (uint32_t*)(geom + GL.tiles_touched);
    uint32_t* slot_base = (uint32_t*)(geom + GL.slot_base);

    g_last_zero_ptr = tile_count; g_last_zero_words = (size_t)2 * T + 64;
    if (!prezeroed) {
        const size_t n_used = (is_used && P > 0) ? (size_t)P : 0;
        size_t blocks = (n_used + (size_t)2 * T + 64 + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tile_count, (size_t)2 * T + 64, (uint32_t*)is_used, n_used);
    }

    int num_rendered = 0;
    if (P > 0) {
        PreprocessArgs pa;
        pa.P = P; pa.D = D; pa.M = M; pa.W = width; pa.H = height; pa.live_rows = live_rows; pa.raw_params = raw_params & 1;
        pa.means3D = means3D; pa.shs = shs; pa.colors_precomp = colors_precomp; pa.opacities = opacities; pa.scales = scales;
        pa.rotations = rotations; pa.cov3D_precomp = cov3D_precomp; pa.scale_modifier = scale_modifier;
        pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos; pa.tanfovx = tan_fovx; pa.tanfovy = tan_fovy;
        pa.tile_mod = tile_mod; pa.tile_rem = tile_rem;
        pa.rec = rec; pa.clamped = (unsigned char*)(geom + GL.clamped);
        pa.tiles_touched = tiles_touched; pa.slot_base = slot_base; pa.total_counter = total_counter;
        pa.vis_list = (uint32_t*)(geom + GL.vis_list); pa.vis_counter = total_counter + 1;
        pa.radii = radii;
        pa.is_used_zero = prezeroed ? is_used : nullptr;
        { ProfileScope ps(ST_PREPROCESS, stream); launch_preprocess(pa, stream); }
        if (async) {
            num_rendered = capacity;   // buffer layouts and launch grids are sized by the capacity; kernels read the true R
        } else {
            uint32_t total = 0;
            GS_CHECK(hipMemcpyAsync(&total, total_counter, 4, hipMemcpyDeviceToHost, stream));
            GS_CHECK(hipStreamSynchronize(stream));   // the one host sync of the forward: the binning buffer is sized by it
            if (total > ID_MASK) { g_last_error = "more than 2^28 (Gaussian, tile) duplicates are not supported"; return -2; }
            num_rendered = (int)total;
        }
        if (num_rendered_dev && !async) hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(1), 0, stream, total_counter, num_rendered_dev);
    } else {
        GS_CHECK(hipStreamSynchronize(stream));
        if (num_rendered_dev) hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)nullptr, num_rendered_dev);
    }
    const uint32_t cap = (uint32_t)num_rendered;

    const BinLayout BL = bin_layout((size_t)num_rendered, (size_t)T);
    char* bin = binning_alloc(binning_user, BL.total);
    if (!bin) { g_last_error = "binning resize callback returned NULL"; return -3; }
    uint32_t* point_list = (uint32_t*)(bin + BL.point_list);
    uint32_t* entry_gauss = (uint32_t*)(bin + BL.entry_gauss);

    int nb = (num_rendered + 4095) / 4096;            // split blocks: >= 4096 slots each, at most SPLIT_BLOCKS_MAX
    if (nb < 1) nb = 1;
    if (nb > SPLIT_BLOCKS_MAX) nb = SPLIT_BLOCKS_MAX;
    const size_t lds_bytes = (size_t)T * 4;
    if (lds_bytes > 160 * 1024) { g_last_error = "image has too many tiles for the LDS tile histogram (> 40 960)"; return -2; }
    if (lds_bytes > 48 * 1024) {   // gfx950 has 160 KiB of LDS per CU, but more than the default dynamic quota must be requested
        static std::atomic<int> raised{0};
        if (!raised.exchange(1)) {
            GS_CHECK(hipFuncSetAttribute((const void*)split_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GS_CHECK(hipFuncSetAttribute((const void*)split_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    uint32_t* emit_tile = (uint32_t*)(bin + BL.emit_tile);
    uint32_t* emit_depth = (uint32_t*)(bin + BL.emit_depth);
    uint32_t* entry_bits = (uint32_t*)(bin + BL.entry_bits);
    uint32_t* block_hist = (uint32_t*)(bin + BL.block_hist);
    if (num_rendered > 0) {
        { ProfileScope ps(ST_EMIT, stream);
          static const bool emit_walk = [] { const char* e = getenv("GSICP_EMIT_WALK"); return e && e[0] == '1'; }();   // A/B: the per-thread walk of rounds 1-4
          if (tile_mod == 1 && !emit_walk)
              hipLaunchKernelGGL(emit_flat_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, total_counter, cap, tiles_touched, slot_base, rec, radii, gx,
                                 gy, emit_tile, emit_depth, entry_gauss, entry_bits, num_rendered_dev);
          else
              hipLaunchKernelGGL(emit_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, total_counter, cap, tiles_touched, slot_base, rec, radii, gx, gy,
                                 tile_mod, tile_rem, emit_tile, emit_depth, entry_gauss, entry_bits, num_rendered_dev); }
        { ProfileScope ps(ST_SPLIT_HIST, stream);
          hipLaunchKernelGGL(split_hist_kernel, dim3(nb), dim3(1024), lds_bytes, stream, total_counter, cap, T, emit_tile, block_hist); }
        // GSICP_TILE_SCAN_MERGED=1: ranges + LPT order by the last workgroup of the column scan instead of a launch of their own — measured and LOST
        // (round 5: 21.9 us against 7.4 + 7.6: the agent-scope fence each of the 13 workgroups needs before it counts itself in is a write-back of its
        // XCD's whole L2, which at that point holds the emit and histogram kernels' fresh output); the default keeps the separate launch
        static const bool scan_launch = [] { const char* e = getenv("GSICP_TILE_SCAN_MERGED"); return !(e && e[0] == '1'); }();
        { ProfileScope ps(ST_SPLIT_COLSCAN, stream);
          hipLaunchKernelGGL(split_colscan_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, T, nb, block_hist, tile_count,
                             scan_launch ? (uint32_t*)nullptr : total_counter + 2, gx, tile_mod, tile_rem, ranges, order); }
        if (scan_launch) {
            ProfileScope ps(ST_RANGES, stream);
            hipLaunchKernelGGL(tile_scan_lpt_kernel, dim3(1), dim3(1024), 0, stream, T, gx, tile_mod, tile_rem, tile_count, ranges, order);
        }
    } else {    // no duplicates at all (synchronous path only): the tile totals are the zeros of zero_fill_kernel
        ProfileScope ps(ST_RANGES, stream);
        hipLaunchKernelGGL(tile_scan_lpt_kernel, dim3(1), dim3(1024), 0, stream, T, gx, tile_mod, tile_rem, tile_count, ranges, order);
    }
    if (num_rendered > 0) {
        {
            ProfileScope ps(ST_SPLIT_SCATTER, stream);
            hipLaunchKernelGGL(split_scatter_kernel, dim3(nb), dim3(1024), lds_bytes, stream, total_counter, cap, T, emit_tile, emit_depth,
                               entry_bits, block_hist, ranges, entry_gauss, (uint4*)(bin + BL.scatter_pack));
        }
        const int n_local = count_local_tiles(gx, gy, tile_mod, tile_rem);
        // ONE class: lists up to SORT_SMALL entries (everything the BASELINE scenes produce: their longest lists are ~500 entries) are sorted
        // in LDS, longer ones in SORT_SMALL-entry chunks merged by rank.  A second, persistent 512-thread kernel for the long lists cost a
        // 4.5 us launch in EVERY iteration to find nothing to do (kernels in a replayed graph cost ~4 us each whatever they compute).
        { ProfileScope ps(ST_TILE_SORT, stream);
          hipLaunchKernelGGL((tile_sort_kernel<SORT_SMALL, 256, 0>), dim3(n_local), dim3(256), 0, stream, n_local, (const uint32_t*)nullptr, order, ranges,
                             (const uint4*)(bin + BL.scatter_pack), point_list,
                             (uint32_t*)(bin + BL.tile_keys), (uint32_t*)(bin + BL.list_gauss), (const SplatRec*)rec, gx, tile_sort_lds_flag().load()); }
    }

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem; ba.depth_mode = depth_mode;
    ba.n_tiles_local = count_local_tiles(gx, gy, tile_mod, tile_rem);
    ba.ranges = ranges; ba.point_list = point_list; ba.rec = rec; ba.list_gauss = (const uint32_t*)(bin + BL.list_gauss);
    ba.order = order;
    ba.bg = background;
    ba.out_color = out_color; ba.out_depth = out_depth; ba.final_T = final_T; ba.n_contrib = n_contrib; ba.is_used = is_used;
    if (ba.n_tiles_local > 0) {
        ProfileScope ps(ST_BLEND_FWD, stream);
        hipLaunchKernelGGL(blend_forward_strip_kernel, dim3(ba.n_tiles_local * 4), dim3(64), 0, stream, ba);
    }
    GS_CHECK(hipGetLastError());
    return num_rendered;
}

int gsicp_raster_forward(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                         gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                         int height, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                         int* is_used, int tile_mod, int tile_rem, int debug, int depth_mode, void* stream) {
    return raster_forward_impl(geom_alloc, geom_user, binning_alloc, binning_user, img_alloc, img_user, P, D, M, background, width, height,
                               means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                               projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, radii, is_used, tile_mod,
                               tile_rem, debug, depth_mode, 0, nullptr, nullptr, 0, stream);
}

int gsicp_raster_forward_async(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                               gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background, int width,
                               int height, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                               float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii,
                               int* is_used, int tile_mod, int tile_rem, int debug, int depth_mode, int capacity,
                               unsigned int* num_rendered_dev, const int* live_rows_dev, int raw_params, void* stream) {
    if (capacity <= 0) { g_last_error = "gsicp_raster_forward_async: capacity must be positive"; return -2; }
    return raster_forward_impl(geom_alloc, geom_user, binning_alloc, binning_user, img_alloc, img_user, P, D, M, background, width, height,
                               means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                               projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, radii, is_used, tile_mod,
                               tile_rem, debug, depth_mode, capacity, num_rendered_dev, live_rows_dev, raw_params, stream);
}

size_t gsicp_raster_backward_scratch_bytes(int num_rendered, int width, int height) {
    (void)width; (void)height;
    const size_t R = num_rendered > 0 ? (size_t)num_rendered : 1;
    return align_up(R * SLOT_F * sizeof(float));
}

int gsicp_raster_backward(int P, int D, int M, int num_rendered, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                          const char* geom_buffer, const char* binning_buffer, const char* img_buffer, char* scratch,
                          const float* dL_dpix, const float* dL_ddepth, float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity,
                          float* dL_dcolors, float* dL_ddepths, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscales, float* dL_drots, int tile_mod, int tile_rem, int debug, int depth_mode, const float* out_depth,
                          const int* live_rows_dev, int raw_params, void* stream_v) {
    (void)debug;
    if (depth_mode < 0 || depth_mode > 1 || (depth_mode == 1 && dL_ddepth && !out_depth)) {
        g_last_error = "gsicp_raster_backward: depth_mode 1 needs the forward's depth image"; return -2;
    }
    hipStream_t stream = (hipStream_t)stream_v;
    if (P <= 0) return 0;
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "bad tile_mod / tile_rem"; return -2; }
    if (!scratch) { g_last_error = "gsicp_raster_backward: scratch buffer is NULL (size it with gsicp_raster_backward_scratch_bytes)"; return -2; }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE, T = gx * gy;
    const GeomLayout GL = geom_layout(P);
    const BinLayout BL = bin_layout((size_t)num_rendered, (size_t)T);
    const ImgLayout IL = img_layout(width, height);
    // num_rendered is the list capacity the forward ran with (== R on the synchronous path); the true R is on the device
    const uint32_t* total_counter = (const uint32_t*)(img_buffer + IL.tile_count) + 2 * (size_t)T;
    float* entry_sum = (float*)scratch;   // (R, SLOT_F): one gradient record per emission slot

    BlendArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.W = width; ba.H = height; ba.gx = gx; ba.tile_mod = tile_mod; ba.tile_rem = tile_rem;
    ba.n_tiles_local = count_local_tiles(gx, gy, tile_mod, tile_rem);
    ba.ranges = (const uint2*)(img_buffer + IL.ranges);
    ba.order = (const uint32_t*)(img_buffer + IL.order);
    ba.point_list = (const uint32_t*)(binning_buffer + BL.point_list);
    ba.list_gauss = (const uint32_t*)(binning_buffer + BL.list_gauss);
    ba.rec = (const SplatRec*)(geom_buffer + GL.records);
    ba.bg = background;
    ba.final_T = (float*)(img_buffer + IL.final_T);
    ba.n_contrib = (uint32_t*)(img_buffer + IL.n_contrib);
    ba.dL_dpix = dL_dpix; ba.dL_ddepth = dL_ddepth; ba.depth_mode = dL_ddepth ? depth_mode : 0; ba.depth_out = out_depth;
    ba.entry_sum = entry_sum;
    if (num_rendered > 0 && ba.n_tiles_local > 0) {
        ProfileScope ps(ST_BLEND_BWD, stream);
        hipLaunchKernelGGL(blend_backward_tile_kernel, dim3(ba.n_tiles_local), dim3(256), 0, stream, ba);
    }

    PreprocessBwdArgs pb;
    pb.P = P; pb.D = D; pb.M = M; pb.W = width; pb.H = height; pb.live_rows = live_rows_dev; pb.raw_params = raw_params & 1;
    pb.sparse_grads = (raw_params >> 1) & 1;
    pb.means3D = means3D; pb.shs = shs; pb.colors_precomp = colors_precomp; pb.scales = scales; pb.rotations = rotations;
    pb.cov3D_precomp = cov3D_precomp; pb.scale_modifier = scale_modifier; pb.view = viewmatrix; pb.proj = projmatrix;
    pb.campos = cam_pos; pb.tanfovx = tan_fovx; pb.tanfovy = tan_fovy; pb.radii = radii;
    pb.clamped = (const unsigned char*)(geom_buffer + GL.clamped);
    pb.entry_sum = entry_sum;
    pb.rec = (const SplatRec*)(geom_buffer + GL.records);
    pb.slot_base = (const uint32_t*)(geom_buffer + GL.slot_base);
    pb.tiles_touched = (const uint32_t*)(geom_buffer + GL.tiles_touched);
    pb.total_counter = total_counter; pb.capacity = (uint32_t)(num_rendered > 0 ? num_rendered : 0);
    pb.entry_gauss = (const uint32_t*)(binning_buffer + BL.entry_gauss);
    pb.vis_list = (const uint32_t*)(geom_buffer + GL.vis_list); pb.vis_counter = total_counter + 1;
    pb.entry_sum_rw = entry_sum;
    pb.dL_dmean2D = dL_dmeans2D; pb.dL_dconic = dL_dconic; pb.dL_dopacity = dL_dopacity; pb.dL_dcolors = dL_dcolors;
    pb.dL_ddepths = dL_ddepths;
    pb.dL_dmeans3D = dL_dmeans3D; pb.dL_dcov3D = dL_dcov3D; pb.dL_dsh = dL_dsh; pb.dL_dscales = dL_dscales; pb.dL_drots = dL_drots;
    { ProfileScope ps(ST_RUN_SUM, stream); launch_entry_run_sum(pb, stream); }
    { ProfileScope ps(ST_PREPROCESS_BWD, stream); launch_preprocess_backward(pb, stream); }
    GS_CHECK(hipGetLastError());
    return 0;
}

int gsicp_raster_set_legacy_backward(int legacy) { return set_prebwd_legacy(legacy); }

int gsicp_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                              unsigned char* present, void* stream) {
    (void)projmatrix;
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    GS_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
