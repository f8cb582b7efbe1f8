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
// alpha >= 1/255 footprint (a conservative bounding box of it) can touch rows 4s..4s+3 of the tile, i.e. the 16x4
// pixel strip that wave s of the blend workgroup owns.  The blend kernels skip entries whose bit is clear.
constexpr uint32_t ID_MASK = 0x0FFFFFFFu;
constexpr int STRIP_SHIFT = 28;
constexpr int NGRAD = 10;    // mean2D x,y | conic a,b,c | opacity | colour r,g,b | depth
constexpr int SLOT_F = 12;   // floats per (entry, strip) gradient slot: NGRAD + 2 pad = three float4

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Section offsets inside the three torch-owned scratch buffers.  Sections needed by backward come first.
struct GeomLayout {
    size_t records, clamped, offsets, ids_sorted, tiles_touched, depth_keys, depth_keys_sorted, ids, scalars, temp, total;
    size_t temp_bytes;
};
struct BinLayout {
    size_t point_list, tile_keys, entry_gauss, entry_pos, point_list_unsorted, tile_keys_unsorted, temp, total;
    size_t temp_bytes;
};
struct ImgLayout {
    size_t ranges, final_T, n_contrib, order, order_keys, order_tmp_keys, order_tmp_vals, sort_temp, total;
    size_t sort_temp_bytes;
};

inline GeomLayout geom_layout(int P, size_t temp_bytes) {
    GeomLayout L;
    size_t o = 0;
    const size_t Pp = (size_t)(P > 0 ? P : 1);
    L.records = o; o = align_up(o + Pp * sizeof(SplatRec));
    L.clamped = o; o = align_up(o + Pp);
    L.offsets = o; o = align_up(o + Pp * 4);          // inclusive scan of tiles_touched in depth-sorted order
    L.ids_sorted = o; o = align_up(o + Pp * 4);       // Gaussian ids in depth order (needed again by backward)
    L.tiles_touched = o; o = align_up(o + Pp * 4);
    L.depth_keys = o; o = align_up(o + Pp * 4);
    L.depth_keys_sorted = o; o = align_up(o + Pp * 4);
    L.ids = o; o = align_up(o + Pp * 4);
    L.scalars = o; o = align_up(o + 256);
    L.temp = o; L.temp_bytes = temp_bytes; o = align_up(o + temp_bytes);
    L.total = o;
    return L;
}
inline BinLayout bin_layout(size_t R, size_t temp_bytes) {
    BinLayout L;
    size_t o = 0;
    const size_t Rp = R > 0 ? R : 1;
    L.point_list = o; o = align_up(o + Rp * 4);
    L.tile_keys = o; o = align_up(o + Rp * 4);
    L.entry_gauss = o; o = align_up(o + Rp * 4);      // Gaussian id of emission slot u (emission order = depth order)
    L.entry_pos = o; o = align_up(o + Rp * 4);        // entry_pos[u] = position of emission slot u in the tile-sorted list
    L.point_list_unsorted = o; o = align_up(o + Rp * 4);
    L.tile_keys_unsorted = o; o = align_up(o + Rp * 4);
    L.temp = o; L.temp_bytes = temp_bytes; o = align_up(o + temp_bytes);
    L.total = o;
    return L;
}
inline ImgLayout img_layout(int W, int H, size_t sort_temp_bytes = 0) {
    ImgLayout L;
    const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const size_t HW = (size_t)W * H;
    size_t o = 0;
    L.ranges = o; o = align_up(o + T * 8);
    L.final_T = o; o = align_up(o + HW * 4);
    L.n_contrib = o; o = align_up(o + HW * 4);
    L.order = o; o = align_up(o + T * 4);            // tiles sorted by list length, longest first (LPT dispatch order)
    L.order_keys = o; o = align_up(o + T * 4);
    L.order_tmp_keys = o; o = align_up(o + T * 4);
    L.order_tmp_vals = o; o = align_up(o + T * 4);
    L.sort_temp = o; L.sort_temp_bytes = sort_temp_bytes; o = align_up(o + sort_temp_bytes);
    L.total = o;
    return L;
}

struct PreprocessArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    float scale_modifier;
    const float *view, *proj, *campos;
    float tanfovx, tanfovy;
    int tile_mod, tile_rem;
    SplatRec* rec;
    unsigned char* clamped;
    uint32_t* tiles_touched;
    uint32_t* depth_keys;
    uint32_t* ids;
    int* radii;
};
static_assert(sizeof(SplatRec) == 48, "SplatRec must stay 48 bytes");

struct PreprocessBwdArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *colors_precomp, *scales, *rotations, *cov3D_precomp;
    float scale_modifier;
    const float *view, *proj, *campos;
    float tanfovx, tanfovy;
    const int* radii;
    const unsigned char* clamped;
    const float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolors, *dL_ddepths;   // (P,3) (P,4) (P) (P,3) (P)
    float *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
};

// ---- per-stage hipEvent profiler (implemented in raster.hip, shared with gicp.hip)
enum Stage { ST_PREPROCESS = 0, ST_DEPTH_SORT, ST_SCAN, ST_DUPLICATE, ST_TILE_SORT, ST_RANGES, ST_BLEND_FWD, ST_BLEND_BWD,
             ST_PREPROCESS_BWD, ST_MEMSET, ST_GICP_COV, ST_GICP_GRID, ST_GICP_ALIGN, ST_GICP_MISS, ST_COUNT };
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
