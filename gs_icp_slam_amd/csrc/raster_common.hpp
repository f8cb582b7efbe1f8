// Shared declarations of the gfx950 rasteriser (not a public header; the public C ABI is include/gsicp_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gsicp {

constexpr int TILE = 16;           // 16x16-pixel tiles (witnessed at REF SIBR_viewers/.../GaussianView.cpp:497-525)
constexpr int TILE_PIX = TILE * TILE;

// One splat's screen-space record, 48 B, written once by preprocess and gathered by the blend kernels
// with three 16-byte loads (one 64-B line touched per gather in the common aligned case).
struct __attribute__((aligned(16))) SplatRec {
    float px, py, depth, hx;       // pixel centre, view-space z, x half-extent of the alpha >= 1/255 footprint
    float ca, cb, cc, opacity;     // conic + opacity
    float r, g, b, hy;             // colour, y half-extent of the footprint
};

// point_list entries: EMISSION SLOT u (index into entry_gauss) in the low 28 bits, 4 "strip" bits on top.  Bit 28+s is set iff the Gaussian's
// alpha >= 1/255 footprint (a conservative bounding box of it) can touch the 8x8-pixel BLOCK s of the tile (s & 1: right half, s >> 1: lower
// half) — the block that wave s of the blend workgroup owns (rounds 1-2: 16x4 strips; square blocks are cut by 6-8 % fewer footprints).
// The blend kernels skip entries whose bit is clear.
constexpr uint32_t ID_MASK = 0x0FFFFFFFu;
constexpr int STRIP_SHIFT = 28;
constexpr int NGRAD = 10;    // mean2D x,y | conic a,b,c | opacity | colour r,g,b | depth
constexpr int SLOT_F = 12;   // floats per emission-slot gradient record: NGRAD + 2 pad = three float4
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Multi-GPU tile ownership (tile_mod = number of ranks, tile_rem = this rank): tiles are dealt round-robin in 2x2 GROUPS — super-tile
// S = (ty / 2) * ceil(gx / 2) + tx / 2 belongs to rank S % tile_mod.  A super-tile is exactly one 32x32-pixel block of the loss kernels, so
// the rank that blends a block also owns its loss gradient (round 3; rounds 1-2 dealt single tiles, t % tile_mod, and every rank computed the
// whole loss).  Interleaving keeps the load balanced; tile_mod <= 1 means "all tiles".
// BAND ownership (round 5): instead of dealing super-tiles round-robin, a rank may own a CONTIGUOUS band of super-tile rows (= rows of 32x32 loss
// blocks).  Encoded in the same two integers so that no signature changes: tile_mod = TILE_BAND_FLAG | hi << 15 | lo owns the super-tile rows
// [lo, hi); tile_rem = world << 16 | rank.  With bands a rank's loss blocks need the image of its two neighbours only within the SSIM window's
// reach (10 pixel rows each way: 0.4 MB at 1200 px instead of the whole 13.7 MB image) — gs_icp_slam_amd/sharded.py, `bands=`.
constexpr int TILE_BAND_FLAG = 1 << 30;
__host__ __device__ inline bool tile_mod_is_band(int tile_mod) { return (tile_mod & TILE_BAND_FLAG) != 0; }
__host__ __device__ inline int tile_world(int tile_mod, int tile_rem) { return tile_mod_is_band(tile_mod) ? (tile_rem >> 16) : tile_mod; }
__host__ __device__ inline bool super_row_is_mine(int srow, int tile_mod) { return srow >= (tile_mod & 0x7FFF) && srow < ((tile_mod >> 15) & 0x7FFF); }
__host__ __device__ inline bool tile_xy_is_mine(int tx, int ty, int gx, int tile_mod, int tile_rem) {
    if (tile_mod_is_band(tile_mod)) return super_row_is_mine(ty >> 1, tile_mod);
    return tile_mod <= 1 || ((((ty >> 1) * ((gx + 1) >> 1)) + (tx >> 1)) % tile_mod) == tile_rem;
}
__host__ __device__ inline bool tile_is_mine(int t, int gx, int tile_mod, int tile_rem) {
    return tile_mod <= 1 || tile_xy_is_mine(t % gx, t / gx, gx, tile_mod, tile_rem);
}
inline int count_local_tiles(int gx, int gy, int tile_mod, int tile_rem) {
    if (tile_mod <= 1) return gx * gy;
    int n = 0;
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) n += tile_xy_is_mine(tx, ty, gx, tile_mod, tile_rem) ? 1 : 0;
    return n;
}
// all-gather chunk of the tile movers: slot k of rank r holds tile (sub = k & 3) of its (k >> 2)-th super-tile; slots past the image are padding
__host__ __device__ inline int tile_chunk_slots(int gx, int gy, int tile_mod) {
    const int n_super = ((gx + 1) >> 1) * ((gy + 1) >> 1);
    return 4 * ((n_super + tile_mod - 1) / tile_mod);
}

// Section offsets inside the three torch-owned scratch buffers.
struct GeomLayout { size_t records, clamped, slot_base, tiles_touched, vis_list, total; };
struct BinLayout { size_t point_list, tile_keys, list_gauss, entry_gauss, entry_bits, emit_tile, emit_depth, scatter_pack, block_hist, total; };
constexpr int SPLIT_BLOCKS_MAX = 128;   // workgroups of the tile multi-split (each owns a contiguous chunk of emission slots)
struct ImgLayout { size_t ranges, final_T, n_contrib, order, tile_count, total; };

inline GeomLayout geom_layout(int P) {
    GeomLayout L;
    size_t o = 0;
    const size_t Pp = (size_t)(P > 0 ? P : 1);
    L.records = o; o = align_up(o + Pp * sizeof(SplatRec));
    L.clamped = o; o = align_up(o + Pp);
    L.slot_base = o; o = align_up(o + Pp * 4);        // first emission slot of each Gaussian (its slots are contiguous)
    L.tiles_touched = o; o = align_up(o + Pp * 4);    // number of emission slots (tiles of this rank it touches)
    L.vis_list = o; o = align_up(o + Pp * 4);         // ids of the Gaussians with radii > 0 (work list of the backward's per-Gaussian pass)
    L.total = o;
    return L;
}
inline BinLayout bin_layout(size_t R, size_t T) {
    BinLayout L;
    size_t o = 0;
    const size_t Rp = R > 0 ? R : 1;
    L.point_list = o; o = align_up(o + Rp * 4);       // per tile, depth-sorted: emission slot | strip bits << 28
    L.tile_keys = o; o = align_up(o + Rp * 4);        // tile id of every list entry
    L.list_gauss = o; o = align_up(o + Rp * 4);       // Gaussian id of every list entry (same order as point_list: coalesced for the blend kernels)
    L.entry_gauss = o; o = align_up(o + Rp * 4);      // emission slot -> Gaussian id
    L.entry_bits = o; o = align_up(o + Rp * 4);       // emission slot -> strip bits
    L.emit_tile = o; o = align_up(o + Rp * 4);        // forward-only: tile id of each emission slot
    L.emit_depth = o; o = align_up(o + Rp * 4);       // forward-only: depth bits of each emission slot
    L.scatter_pack = o; o = align_up(o + Rp * 16);    // forward-only: {depth bits, list word, Gaussian id, -} per entry in scatter order (ONE 16-byte
                                                      // store per entry; the per-tile sort reads it coalesced and gathers nothing)
    L.block_hist = o; o = align_up(o + (size_t)SPLIT_BLOCKS_MAX * (T > 0 ? T : 1) * 4);   // forward-only: per-(split block, tile) counts
    L.total = o;
    return L;
}
inline ImgLayout img_layout(int W, int H) {
    ImgLayout L;
    const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const size_t HW = (size_t)W * H;
    size_t o = 0;
    L.ranges = o; o = align_up(o + T * 8);
    L.final_T = o; o = align_up(o + HW * 4);
    L.n_contrib = o; o = align_up(o + HW * 4);
    L.order = o; o = align_up(o + T * 4);             // this rank's tiles, longest list first (LPT dispatch order)
    L.tile_count = o; o = align_up(o + (2 * T + 64) * 4);   // per-tile counts, cursors, 64 counters ([0] = R)
    L.total = o;
    return L;
}

struct PreprocessArgs {
    int P, D, M, W, H;
    const int* live_rows;      // DEVICE, optional: only rows [0, min(P, *live_rows)) are Gaussians (capacity-backed map, captured graphs)
    int raw_params;            // 1: opacities / scales / rotations are GaussianModel's RAW parameters; sigmoid / exp / normalize happen here
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    float scale_modifier;
    const float *view, *proj, *campos;
    float tanfovx, tanfovy;
    int tile_mod, tile_rem;
    SplatRec* rec;
    unsigned char* clamped;
    uint32_t* tiles_touched;
    uint32_t* slot_base;
    uint32_t* total_counter;   // [1] emission-slot allocator
    uint32_t* vis_list;        // optional: ids of the Gaussians with radii > 0, compacted ...
    uint32_t* vis_counter;     // ... and their number (zeroed with total_counter)
    int* radii;
    int* is_used_zero;         // optional (pre-zeroed forward, round 6): thread i clears is_used[i] here — the blend kernel sets it later —, so that the forward needs no
                               // zero-fill launch when its per-call counters were cleared ahead of it (gsicp_mapper_select_view_zero)
};
static_assert(sizeof(SplatRec) == 48, "SplatRec must stay 48 bytes");

struct PreprocessBwdArgs {
    int P, D, M, W, H;
    const int* live_rows;      // DEVICE, optional (see PreprocessArgs)
    int raw_params;            // 1: scales / rotations are raw and dL_dopacity / dL_dscales / dL_drots are gradients w.r.t. the RAW parameters
    int sparse_grads;          // 1 (round 6): the gradient rows of CULLED Gaussians (radii == 0, or behind the live count) are NOT written — their value is zero
                               //    by definition and the consumer takes it from radii (FusedAdam's row mask); 0: every row of every output is written
    const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp;
    float scale_modifier;
    const float *view, *proj, *campos;
    float tanfovx, tanfovy;
    const int* radii;
    const unsigned char* clamped;
    const float* entry_sum;        // (R, SLOT_F) per-emission-slot moment sums [h dx, h dy, h dx dx, h dx dy, h dy dy, h, w r, w g, w b, w z]
    const SplatRec* rec;           // conic + opacity of every Gaussian (the per-entry constants of the blend backward are applied here)
    const uint32_t *slot_base, *tiles_touched;
    const uint32_t* total_counter;  // device R; above `capacity` the forward rendered nothing (async path overflow)
    uint32_t capacity;
    const uint32_t* entry_gauss;    // emission slot -> Gaussian id (the run-sum pass finds the run boundaries in it)
    const uint32_t *vis_list, *vis_counter;   // the forward's work list (ids with radii > 0) and its length
    float* entry_sum_rw;            // = entry_sum: the run-sum pass leaves a run's total in the record of its LAST slot
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolors, *dL_ddepths;   // (P,3) (P,4) (P) (P,3) (P): written here
    float *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
};

// ---- per-stage hipEvent profiler (implemented in raster.hip, shared with gicp.hip)
// one stage per KERNEL on the rasteriser side (a bracket over several launches would also time the host gaps between them)
enum Stage { ST_PREPROCESS = 0, ST_RANGES, ST_EMIT, ST_SPLIT_HIST, ST_SPLIT_COLSCAN, ST_SPLIT_SCATTER, ST_TILE_SORT, ST_BLEND_FWD,
             ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_GICP_COV, ST_GICP_GRID, ST_GICP_ALIGN, ST_GICP_MISS, ST_LOSS_PASS1, ST_LOSS_PASS2, ST_ADAM, ST_RUN_SUM, ST_COUNT };
bool profile_on();
void profile_begin(int stage, hipStream_t s);
void profile_end(int stage, hipStream_t s);
struct ProfileScope {
    int st; hipStream_t s; bool on;
    ProfileScope(int stage, hipStream_t stream) : st(stage), s(stream), on(profile_on()) { if (on) profile_begin(st, s); }
    ~ProfileScope() { if (on) profile_end(st, s); }
};

// implemented in raster_preprocess.hip (compiled with -ffp-contract=off so integer-feeding floats are reproducible)
void launch_preprocess(const PreprocessArgs& a, hipStream_t s);
void launch_preprocess_backward(const PreprocessBwdArgs& a, hipStream_t s);
void launch_entry_run_sum(const PreprocessBwdArgs& a, hipStream_t s);
int set_prebwd_legacy(int legacy);   // returns the previous setting
void launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s);

// Tile rectangle of a splat (shared by preprocess and the duplicate kernel; integer outputs must agree).
__host__ __device__ inline void tile_rect(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
    const float r = (float)rad;
    int a;
    a = (int)((px - r) / (float)TILE); x0 = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py - r) / (float)TILE); y0 = a < 0 ? 0 : (a > gy ? gy : a);
    a = (int)((px + r + (float)(TILE - 1)) / (float)TILE); x1 = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py + r + (float)(TILE - 1)) / (float)TILE); y1 = a < 0 ? 0 : (a > gy ? gy : a);
}

}  // namespace gsicp
