/* gsicp_hip.h — C ABI of libgsicp_hip.so, the MI355X (gfx950) implementation of GS-ICP-SLAM's per-frame hot path.
 *
 * Plain pointers and sizes only: no torch / pybind / Eigen types cross this boundary.  Every DEVICE pointer is a
 * HIP device address valid on the current device; `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  All work is enqueued on `stream`; functions documented "synchronous" wait for it before returning.
 * Return value: >= 0 on success (meaning given per function), < 0 on failure; gsicp_last_error() then returns
 * a thread-local message.  The reference side of each entry point (what a maintainer binds it to) is cited as
 * [REF file:line] relative to /root/reference; INTEGRATION.md shows the Python (ctypes) bindings shipped in
 * gs_icp_slam_amd/ and the pybind11 stub equivalent.
 *
 * The reference's own native sources for this path are absent from its tree (empty submodules,
 * [REF .gitmodules:1-10]); the interfaces replaced are therefore identified by their Python call sites.
 */
#ifndef GSICP_HIP_H
#define GSICP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSICP_ABI_VERSION 5

int gsicp_abi_version(void);
const char* gsicp_last_error(void);
/* Number of visible HIP devices (<0: runtime error). */
int gsicp_device_count(void);

/* Per-stage hipEvent timers (off by default).  While enabled, every kernel / library call of the rasteriser and the
 * GICP align kernel is bracketed by an event pair on the stream it is launched on.  gsicp_profile_read() synchronises
 * the recorded events, writes the accumulated milliseconds and launch counts per stage (order = stage index) and
 * clears the log; it returns the number of stages.  This is what bench.py's `roofline` block is measured with. */
int gsicp_profile_enable(int on);
int gsicp_profile_num_stages(void);
const char* gsicp_profile_stage_name(int stage);
int gsicp_profile_read(double* ms_out, int* count_out, int capacity);

/* --------------------------------------------------------------------------------------------------------
 * 1. Rasteriser — replaces diff_gaussian_rasterization._C.{rasterize_gaussians, rasterize_gaussians_backward,
 *    mark_visible}, reached from GaussianRasterizer.forward at [REF gaussian_renderer/__init__.py:294-302] and
 *    from loss.backward() at [REF mp_Mapper.py:242].
 * ------------------------------------------------------------------------------------------------------ */

/* Scratch-buffer resize callback (same convention the viewer's C++ caller uses for the upstream rasteriser,
 * [REF SIBR_viewers/src/projects/gaussianviewer/renderer/GaussianView.cpp:304,432-434]): must return a DEVICE
 * pointer to at least `bytes` bytes, 256-byte aligned, that stays alive until the matching backward call. */
typedef char* (*gsicp_resize_fn)(void* user, size_t bytes);

/* Forward: preprocess -> depth sort -> tile binning -> tile sort -> front-to-back blend.
 *   P Gaussians; D = active SH degree (0..3); M = SH coefficients per Gaussian in `shs` (>= (D+1)^2).
 *   All float inputs are contiguous f32 on the device.  Exactly one of shs / colors_precomp and one of
 *   (scales, rotations) / cov3D_precomp must be non-NULL.  rotations are quaternions in (x,y,z,w) order
 *   [REF utils/general_utils.py:89-99]; viewmatrix / projmatrix are the row-vector (pre-transposed) 4x4s of
 *   [REF scene/shared_objs.py:163-166].
 *   Outputs: out_color (3,H,W), out_depth (1,H,W), radii (P) int32, is_used (P) int32.
 *   depth_mode selects the depth compositing rule — one of the three fork semantics the reference tree cannot settle (its rasteriser
 *   fork is an empty submodule): 0 = sum_i z_i alpha_i T_i (default; no normalisation, no background term),
 *   1 = alpha-normalised, sum_i z_i alpha_i T_i / (1 - T_final) (0 where nothing was blended).  Every reported number names the variant.
 *   tile_mod / tile_rem: multi-GPU tile sharding (pass 1, 0 for the whole image).  Tiles are dealt round-robin in 2x2 SUPER-TILES: with
 *   gx = ceil(W / 16), super-tile S = (ty / 2) * ceil(gx / 2) + tx / 2 of tile (tx, ty) belongs to rank S % tile_mod; this call blends only the
 *   tiles of rank tile_rem and leaves the other tiles' pixels untouched.  (ABI version 4; versions <= 3 dealt single tiles, tile_id %
 *   tile_mod.)  A super-tile is one 32x32-pixel block of gsicp_mapper_loss_sharded: the rank that blends a block owns its loss gradient.
 * Returns the number of (Gaussian, tile) duplicates binned for this call (the reference's `num_rendered`).
 * Synchronises `stream` once (to size the binning buffer), like the reference implementation. */
int gsicp_raster_forward(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                         gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background,
                         int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                         float* out_depth, int* radii, int* is_used, int tile_mod, int tile_rem, int debug, int depth_mode,
                         void* stream);

/* Same forward without the host round trip (extension; no reference counterpart — the reference always reads
 * num_rendered back, [REF submodules/diff-gaussian-rasterization: rasterizer_impl forward] behind
 * gaussian_renderer/__init__.py:294): the caller states the duplicate-list `capacity`, every buffer and launch is
 * sized by it and the kernels read the true count R from device memory.  Nothing in the call synchronises, allocates
 * outside the callbacks or copies to the host, so it can be captured in a HIP graph together with the backward.
 *   num_rendered_dev: DEVICE uint32 that receives R (may be NULL).  If R > capacity the call renders nothing
 *   (background colour, zero depth, zero gradients) — the caller detects it by reading num_rendered_dev later and
 *   repeats with a larger capacity.
 *   live_rows_dev: optional DEVICE int.  When non-NULL only rows [0, min(P, *live_rows_dev)) are Gaussians and the rest of the P-row
 *   arrays is ignored (treated as culled; zero gradients): P is then the CAPACITY of a preallocated map whose live count changes on the
 *   device — every pointer and launch grid stays what it was, so a captured graph survives map growth and pruning (pass the same
 *   pointer to gsicp_raster_backward, the activation operators and gsicp_adam_step_guarded).
 *   raw_params: 1 = `opacities`, `scales`, `rotations` are GaussianModel's RAW parameters (logit opacity, log scales, un-normalised
 *   quaternion) and the activation getters [REF scene/gaussian_model.py:44-56, 105-125] — sigmoid, exp, x / max(||x||, 1e-12) — are applied
 *   inside the preprocess kernel; gsicp_raster_backward with the same flag then returns dL_dopacity / dL_dscales / dL_drots with respect
 *   to the raw parameters (two element-wise launches and 64 B per Gaussian of traffic less per iteration).
 *   gsicp_raster_backward only (ABI 5): raw_params bit 1 (value 2) = SPARSE gradients — the rows of culled Gaussians (radii == 0, or behind the
 *   live count) are NOT written to any dL_d* output; their value is zero by definition and the consumer masks by `radii`
 *   (gsicp_adam_step_sparse).  Without the bit every row of every output is written (zeros for culled rows), as the reference's backward does.
 * Returns `capacity`; pass that value as `num_rendered` to gsicp_raster_backward_scratch_bytes / gsicp_raster_backward. */
int gsicp_raster_forward_async(gsicp_resize_fn geom_alloc, void* geom_user, gsicp_resize_fn binning_alloc, void* binning_user,
                               gsicp_resize_fn img_alloc, void* img_user, int P, int D, int M, const float* background,
                               int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                               const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                               float* out_depth, int* radii, int* is_used, int tile_mod, int tile_rem, int debug, int depth_mode,
                               int capacity, unsigned int* num_rendered_dev, const int* live_rows_dev, int raw_params, void* stream);

/* Bytes of DEVICE scratch gsicp_raster_backward needs (one 48-byte gradient record per (Gaussian, tile) duplicate; contents need no
 * initialisation and are dead after the call). */
size_t gsicp_raster_backward_scratch_bytes(int num_rendered, int width, int height);

/* Backward of the call above.  geom/binning/img buffers are the ones the forward call filled; num_rendered its
 * return value; `scratch` is a DEVICE buffer of gsicp_raster_backward_scratch_bytes() bytes.
 * dL_dpix (3,H,W) and dL_ddepth (1,H,W; may be NULL) are the incoming image gradients; depth_mode as in the forward, and with
 * depth_mode 1 `out_depth` must be the depth image that forward produced (may be NULL otherwise).
 * No atomics are used: gradients are bit-reproducible from run to run.
 * Gradient outputs (all DEVICE, all fully overwritten): dL_dmeans2D (P,3) [x,y in NDC-scaled units, z = 0],
 * dL_dopacity (P), dL_dmeans3D (P,3), dL_dsh (P,M,3; may be NULL when colors_precomp is used), dL_dscales (P,3), dL_drots (P,4);
 * intermediate results, each may be NULL (its writes are then skipped): dL_dconic (P,4), dL_ddepths (P), dL_dcolors (P,3) [needed when
 * colors_precomp is the input], dL_dcov3D (P,6) [needed when cov3D_precomp is the input].
 * Asynchronous on `stream`. */
int gsicp_raster_backward(int P, int D, int M, int num_rendered, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                          float tan_fovy, const int* radii, const char* geom_buffer, const char* binning_buffer,
                          const char* img_buffer, char* scratch, const float* dL_dpix, const float* dL_ddepth,
                          float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_ddepths,
                          float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots, int tile_mod,
                          int tile_rem, int debug, int depth_mode, const float* out_depth, const int* live_rows_dev, int raw_params,
                          void* stream);

/* present[i] = 1 iff Gaussian i passes the frustum test (view-space z > 0.2).  Asynchronous. */
int gsicp_raster_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                              unsigned char* present, void* stream);

/* Introspection used by the parity tests: byte offsets of the sections inside the three scratch buffers
 * (so tests can read the sorted (tile, Gaussian) lists and tile ranges bit-exactly).  out[] receives
 * {geom_bytes, binning_bytes, img_bytes, off_records, off_point_list, off_tile_keys, off_ranges, off_final_T,
 *  off_n_contrib, off_clamped, off_entry_gauss, off_entry_pos}.  A point_list word holds an emission-slot index in
 * its low 28 bits (entry_gauss[slot] is the Gaussian id) and 4 strip bits on top. */
int gsicp_raster_layout(int P, int num_rendered, int width, int height, size_t out[12]);

/* Test / A-B hook: which per-Gaussian pass gsicp_raster_backward launches.  0 (default) = round 5's two launches (flat run summation over the
 * emission slots + the algebra over the forward's compacted list of visible Gaussians); 1 = the legacy single kernel of rounds 3-4 (one thread per
 * Gaussian walking its own run).  Process-wide; also set by GSICP_PREBWD_LEGACY=1 at load.  Returns the previous value.  Both compute the
 * same gradients; the order in which a Gaussian's per-tile records are added differs (tests/test_raster_gpu.py compares them). */
int gsicp_raster_set_legacy_backward(int legacy);

/* Test / A-B hook: which sorting network the per-tile sort of the forward runs.  0 (default) = round 6's — every compare-exchange with partner distance < 128 in
 * REGISTERS (lane exchanges by DPP / v_permlane swaps; only the cross-block steps of lists above 128 entries go through LDS); 1 = every step through LDS, as rounds 2-5.
 * Process-wide; also GSICP_TILE_SORT_LDS=1 at load.  Returns the previous value.  Same network, same total order: the lists are the same bits
 * (tests/test_raster_gpu.py). */
int gsicp_raster_set_tile_sort_lds(int lds_only);

/* A HIP stream restricted to `n_cus` compute units starting at CU-mask bit `first_cu` (hipExtStreamCreateWithCUMask; MI355X: 256 CUs in 8 XCDs, consecutive
 * mask bits go round the XCDs, so a contiguous range is spread evenly over them).  Round 6 experiment (default off): the tracker on dedicated CUs
 * (GSICP_TRACKER_CU_MASK="first:count" at gsicp_gicp_create) and the mapper's stream on the complement, instead of a high-priority tracker stream sharing
 * all CUs.  Returns the hipStream_t (NULL on failure); wrap it with torch.cuda.ExternalStream.  No reference counterpart (the reference runs two processes
 * on one GPU and leaves the sharing to the hardware scheduler [REF gs_icp_slam.py:121-131]). */
void* gsicp_stream_create_cu_mask(int first_cu, int n_cus);
int gsicp_stream_destroy(void* stream);

/* --------------------------------------------------------------------------------------------------------
 * 2. simple_knn — replaces simple_knn._C.distCUDA2 [REF scene/gaussian_model.py:20 (import site)].
 *    out[i] = mean squared distance from point i to its 3 nearest other points (f32).  Asynchronous.
 * ------------------------------------------------------------------------------------------------------ */
int gsicp_knn_dist2(int P, const float* points, float* out, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * 3. GICP tracker — replaces pygicp.FastGICP [REF mp_Tracker.py:53].  One opaque object per tracker; all
 *    functions below are synchronous with respect to their HOST arguments (host numpy in, host numpy out, as at
 *    the reference's call sites) and run their kernels on the object's own stream.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct gsicp_gicp gsicp_gicp;

gsicp_gicp* gsicp_gicp_create(void);                       /* pygicp.FastGICP()            [REF mp_Tracker.py:53]  */
void gsicp_gicp_destroy(gsicp_gicp*);
int gsicp_gicp_set_max_correspondence_distance(gsicp_gicp*, double d);     /* [REF mp_Tracker.py:109] */
int gsicp_gicp_set_max_knn_distance(gsicp_gicp*, double d);                /* [REF mp_Tracker.py:110] */
int gsicp_gicp_set_correspondence_randomness(gsicp_gicp*, int k);          /* upstream API; k-NN size, default 20 */
int gsicp_gicp_set_max_iterations(gsicp_gicp*, int n);                     /* upstream API; default 64 */
int gsicp_gicp_set_num_threads(gsicp_gicp*, int n);                        /* upstream API; accepted, ignored */
/* method: 0 NONE, 1 MIN_EIG, 2 NORMALIZED_MIN_EIG, 3 PLANE (default), 4 FROBENIUS */
int gsicp_gicp_set_regularization_method(gsicp_gicp*, int method);
/* One of the three fork semantics that cannot be read off the reference (its fast_gicp fork is an empty submodule): what the
 * numbers of get_*_scales() / set_target_covariances_fromqs() ARE.  0 (default) = standard deviations: exported scales are the square
 * roots of the k-NN covariance eigenvalues and fromqs builds R diag(s^2) R^T — the mapper uses the exported values as 3DGS scales
 * [REF scene/gaussian_model.py:143-145].  1 = variances: eigenvalues are exported as they are and fromqs builds R diag(s) R^T.  Both
 * round-trip a covariance through the map; every reported number names the variant. */
int gsicp_gicp_set_scale_semantics(gsicp_gicp* g, int mode);
int gsicp_gicp_set_rotation_epsilon(gsicp_gicp*, double eps);
int gsicp_gicp_set_transformation_epsilon(gsicp_gicp*, double eps);

/* points: HOST (n,3) row-major, f32 (is_f64 = 0) or f64 (is_f64 = 1)  [REF mp_Tracker.py:157,191,287] */
int gsicp_gicp_set_input_target(gsicp_gicp*, const void* points, int n, int is_f64);
int gsicp_gicp_set_input_source(gsicp_gicp*, const void* points, int n, int is_f64);
/* filter: HOST int32 (n_points); 0 = not trackable, r>0 = r-th trackable point  [REF mp_Tracker.py:159-163,192-195] */
int gsicp_gicp_set_target_filter(gsicp_gicp*, int n_trackable, const int32_t* filter, int n_points);
int gsicp_gicp_set_source_filter(gsicp_gicp*, int n_trackable, const int32_t* filter, int n_points);
int gsicp_gicp_calculate_target_covariance_with_filter(gsicp_gicp*);       /* [REF mp_Tracker.py:164] */
int gsicp_gicp_calculate_source_covariance(gsicp_gicp*);                   /* implicit inside align() upstream */
/* out: HOST f32; rotations 4 per point (x,y,z,w), scales 3 per point (std-devs, descending)
 * [REF mp_Tracker.py:166-169,256-264].  Returns the number of points written. */
int gsicp_gicp_get_target_rotationsq(gsicp_gicp*, float* out, int capacity_points);
int gsicp_gicp_get_target_scales(gsicp_gicp*, float* out, int capacity_points);
int gsicp_gicp_get_source_rotationsq(gsicp_gicp*, float* out, int capacity_points);
int gsicp_gicp_get_source_scales(gsicp_gicp*, float* out, int capacity_points);
/* Sigma_i = R(q_i) diag(s_i^2) R(q_i)^T for the K current target points  [REF mp_Tracker.py:288] */
int gsicp_gicp_set_target_covariances_fromqs(gsicp_gicp*, const float* rots_flat, int n_rots, const float* scales_flat,
                                             int n_scales);
/* initial/final: HOST 4x4 row-major f64 camera-to-world.  Returns the number of outer iterations (>= 0)
 * [REF mp_Tracker.py:199].  Synchronous, like every call of this object.  The FIRST align after a target change also builds the target index
 * (lazily): for targets of >= 32 768 points that build reads the occupied-cell count back once per index level (one or two 4-byte
 * device-to-host reads, each a stream drain) and may re-allocate the hash table and cell records sized by it — expect that align to
 * take 0.3-3 ms longer and not to be allocation-free. */
int gsicp_gicp_align(gsicp_gicp*, const double* initial_pose, double* final_pose);
/* One entry per trackable source point: nearest-target index (-1 when farther than the correspondence gate) and
 * the squared distance to the nearest target point, as of the last linearisation  [REF mp_Tracker.py:231].
 * Returns the number of entries.  The first call on an object computes and exports on demand; from then on gsicp_gicp_align
 * enqueues the kernels behind its own (the reference asks after every align), and this call waits for the export and copies. */
int gsicp_gicp_get_source_correspondence(gsicp_gicp*, int32_t* target_index, float* sq_distance, int capacity);
/* ---- Device-pointer overloads (additive — the numpy-style entry points above are unchanged) ----------
 * The reference hands the map's trackable Gaussians to the tracker through host memory on every tracking keyframe while the
 * tracker blocks: GPU -> CPU tensors -> numpy -> set_input_target + set_target_covariances_fromqs
 * [REF scene/gaussian_model.py:207-215; scene/shared_objs.py:81-126; mp_Tracker.py:284-289].  These take DEVICE pointers
 * (contiguous f32).  `producer_stream` is the stream the buffers were written on: the tracker's stream is ordered after it.
 * With wait = 0 the call returns once the work is enqueued and the caller must keep the buffers alive (and unmodified) until
 * the next synchronous tracker call returns; with wait = 1 the buffers are consumed on return. */
void* gsicp_gicp_stream(gsicp_gicp*);   /* the object's hipStream_t (for record_stream-style lifetime management) */
int gsicp_gicp_set_input_target_device(gsicp_gicp*, const float* xyz /* (n,3) */, int n, void* producer_stream, int wait);
int gsicp_gicp_set_input_source_device(gsicp_gicp*, const float* xyz /* (n,3) */, int n, void* producer_stream, int wait);
int gsicp_gicp_set_target_covariances_fromqs_device(gsicp_gicp*, const float* rots /* (n,4) xyzw */, int n_rots, const float* scales /* (n,3) */,
                                                    int n_scales, void* producer_stream, int wait);
/* get_trackable_gaussians_tensor + set_input_target + set_target_covariances_fromqs in one call: keeps Gaussian i iff
 * opacity[i] > opacity_th and trackable_mask[i] != 0 (mask may be NULL), in index order (what torch's boolean indexing yields),
 * and installs the survivors as the target cloud with covariances R diag(s^2) R^T (+ the configured regularisation).
 * rotation / scaling / opacity are the ACTIVATED values (get_rotation, get_scaling, get_opacity).  Returns the number of target
 * points (synchronises once for that count). */
int gsicp_gicp_set_target_from_gaussians_device(gsicp_gicp*, int P, const float* xyz, const float* rotation, const float* scaling,
                                                const float* opacity, const unsigned char* trackable_mask, float opacity_th,
                                                void* producer_stream);
/* Source covariances as quaternions / scales written into DEVICE buffers; `consumer_stream` is ordered after the copy (no host wait). */
int gsicp_gicp_get_source_rotationsq_device(gsicp_gicp*, float* out_dev, int cap_points, void* consumer_stream);
int gsicp_gicp_get_source_scales_device(gsicp_gicp*, float* out_dev, int cap_points, void* consumer_stream);

/* Diagnostics of the last k-NN covariance pass: out = {cell edge, nx, ny, nz, cells, queries settled by whole-grid coverage,
 * after ring 1, ring 2, ring 3, by the exhaustive scan, 0, 0}.  Synchronises. */
int gsicp_gicp_knn_stats(gsicp_gicp* g, double out[12]);
/* Diagnostics (only with GSICP_ALIGN_TRACE set in the environment): (tag, wall_clock64 at 100 MHz) pairs stamped by workgroup 0 at
 * the phase boundaries of the last align.  Returns the number of pairs written (0 when tracing is off). */
int gsicp_gicp_align_trace(gsicp_gicp* g, unsigned long long* out, int cap_pairs);
int gsicp_gicp_num_source(gsicp_gicp*);
int gsicp_gicp_num_target(gsicp_gicp*);
/* Sizes of the target search structure as of the last build (align builds it lazily after a target / gate change):
 * out[0] = trackable targets, out[1] = uses the hashed grid (0 = ungated scan); then per level l = 0 (complete within the gate) and
 * l = 1 (dense maps only; all zero when absent) at out[2 + 5 l ..]: hash-table slots, bytes of table + cell records, hashed (coarse)
 * cell edge in metres (a fine cell is half of it), occupied coarse cells, radius the level is complete within.  Synchronises. */
int gsicp_gicp_target_index_stats(gsicp_gicp*, double out[12]);
/* out[0..5]: kernel launches of the last align, LM trials, final cost, converged flag, device microseconds, reserved */
int gsicp_gicp_last_align_stats(gsicp_gicp*, double out[6]);
/* Robustness of the persistent align kernel's grid barrier.  The launch never exceeds the number of workgroups the device can hold at
 * once; if a barrier nevertheless times out (a co-tenant kept some workgroup from becoming resident for ~0.2 s) the registration is re-run
 * as ONE workgroup, which needs no grid barrier.  gsicp_gicp_barrier_retries() counts such re-runs; gsicp_gicp_debug_abort_next_align()
 * is a test hook that makes the next align's first barrier abort, so that the recovery path can be exercised deterministically. */
int gsicp_gicp_debug_abort_next_align(gsicp_gicp* g);
/* Test hook: sorts 64 (squared distance >= 0, id >= 0) pairs with the wave-wide network the k-NN kernels use (ascending by distance, then
 * id) and returns, for J in {1, 2, 4, 8, 15, 16, 32}, what lane l reads from lane l ^ J when every lane holds 3 l + 1 (out_xor[7][64]). */
int gsicp_debug_wave_sort(const float* d, const int* id, float* out_d, int* out_id, int* out_xor);
int gsicp_gicp_barrier_retries(gsicp_gicp* g);
int gsicp_gicp_get_final_hessian(gsicp_gicp*, double out[36]);

/* --------------------------------------------------------------------------------------------------------
 * 4. Mapper-side operators next to the rasteriser.  Not extension modules in the reference:
 *    they replace chains of torch ops in its Python, so adopting them means a three-line edit of mp_Mapper.py
 *    (INTEGRATION.md §6).  All pointers DEVICE, contiguous f32; asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------------------ */

/* The mapping loss of [REF mp_Mapper.py:225-240] with the reference's masked L1 and 11x11 Gaussian-window SSIM
 * [REF utils/loss_utils.py:17-20, 27-69]:
 *     m   = gt_depth > 0;  y = gt_image * m;  x = where(y != 0, image, 0)
 *     loss = (1 - lambda) * mean(where(y != 0, |image - y|, 0)) + lambda * (1 - mean(SSIM(x, y)))
 *            + depth_weight * mean(where(gt_depth != 0, |depth - gt_depth| / d_max, 0))
 * image/gt_image (3,H,W), depth/gt_depth (1,H,W).  loss_out[4] = {loss, L1, SSIM mean, depth L1}.
 * dL_dimage (3,H,W) / dL_ddepth (1,H,W) receive d loss / d image, d depth (both may be NULL to skip the gradient pass).
 * scratch: gsicp_mapper_loss_scratch_bytes() bytes, no initialisation needed. */
size_t gsicp_mapper_loss_scratch_bytes(int width, int height);
int gsicp_mapper_loss(const float* image, const float* depth, const float* gt_image, const float* gt_depth, int width, int height,
                      float lambda_dssim, float depth_weight, float d_max, float* loss_out, float* dL_dimage, float* dL_ddepth,
                      char* scratch, void* stream);
/* The same loss SHARDED over tile_mod ranks (multi-GPU mapper, section 4b): rank tile_rem works on the 32x32-pixel blocks it owns — block
 * (bx, by) belongs to rank (by * ceil(W / 32) + bx) % tile_mod, which is exactly the 2x2 super-tile rule by which the rasteriser deals its
 * tiles (tile_mod / tile_rem of gsicp_raster_forward), so the rank that blends a block also owns its loss gradient.  `image` / `depth` must
 * hold the WHOLE rendered image (a block's SSIM windows and their gradient reach 10 pixels into the neighbouring blocks: after
 * gsicp_tiles_unpack).  dL_dimage / dL_ddepth are written on the rank's own blocks only (the other pixels are left untouched and are not read
 * by that rank's backward).  loss_out[4] receives this rank's SHARE of {loss, L1, SSIM mean, depth L1}: the sums over the ranks are the values
 * gsicp_mapper_loss returns (the loss's constant lambda is split tile_mod ways).  One fused kernel + a one-workgroup sum.  tile_mod = 1 is
 * gsicp_mapper_loss. */
int gsicp_mapper_loss_sharded(const float* image, const float* depth, const float* gt_image, const float* gt_depth, int width, int height,
                              float lambda_dssim, float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out,
                              float* dL_dimage, float* dL_ddepth, char* scratch, void* stream);

/* Keyframe selection for a captured (hipGraph) mapper iteration: copies the camera (viewmatrix 16, projmatrix 16, campos 3 floats)
 * and the two target images (gt_image (3,H,W), gt_depth (1,H,W)) of the chosen keyframe [REF mp_Mapper.py:205-217] into the fixed
 * DEVICE buffers the captured kernels read, in one launch.  All pointers are DEVICE pointers; images 16-byte aligned, W*H % 4 == 0. */
int gsicp_mapper_set_view(int width, int height, const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image,
                          const float* gt_depth, float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, float* dst_gt_image,
                          float* dst_gt_depth, void* stream);

/* Round 5: keyframe selection WITHOUT moving the images.  The captured iteration's loss kernels read the two ground-truth pointers from a DEVICE
 * slot pair (gsicp_mapper_loss_indirect below); gsicp_mapper_select_view writes the camera (16 + 16 + 3 floats) into the graph's static buffers and the
 * two pointers into `dst_gt_slots` (device memory, 2 x 8 bytes) in one 64-thread launch.  The images must stay valid (and unchanged) until the
 * replay that reads them has finished — the caller keeps the keyframe's tensors alive, as the reference's mapping_cams list does
 * [REF mp_Mapper.py:147, 175]. */
int gsicp_mapper_select_view(const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image, const float* gt_depth,
                             float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, const float** dst_gt_slots, void* stream);
/* PRE-ZEROED forward (ABI 5, round 6): a captured mapper iteration replays the SAME rasteriser forward with the SAME scratch buffers, so the region the forward clears
 * at its start (per-tile counts and cursors, the slot / visible counters) can be cleared ahead of it by the keyframe-selection launch that precedes every replay anyway:
 *   gsicp_raster_last_zero_region   -> the counter region (pointer, 4-byte words) of this thread's last gsicp_raster_forward{,_async} call; the caller of a CAPTURE
 *                                      reads it right after the captured forward call (the buffers of a hipGraph keep their addresses);
 *   gsicp_mapper_select_view_zero   =  gsicp_mapper_select_view + a clear of that region, one 256-thread launch;
 *   gsicp_raster_forward_async(debug | 2)  skips its zero-fill launch (the preprocess kernel clears `is_used`).  The caller guarantees that the clear is ordered,
 *                                      on the same stream, between the previous use of the buffers and this call.
 * gs_icp_slam_amd/graph.py does all three (15 -> 14 kernel nodes per captured iteration); every other caller keeps the self-contained forward. */
int gsicp_raster_last_zero_region(void** ptr, size_t* words);
int gsicp_mapper_select_view_zero(const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image, const float* gt_depth,
                                  float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, const float** dst_gt_slots, void* zero_region,
                                  size_t zero_words, void* stream);
/* gsicp_mapper_loss / gsicp_mapper_loss_sharded (tile_mod > 1) with the ground-truth images taken from gt_slots[0] (3,H,W) and gt_slots[1] (1,H,W),
 * read ON THE DEVICE when the kernels start. */
int gsicp_mapper_loss_indirect(const float* image, const float* depth, const float* const* gt_slots, int width, int height, float lambda_dssim,
                               float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out, float* dL_dimage, float* dL_ddepth,
                               char* scratch, void* stream);

/* Test / A-B hook: 1 (default) = loss pass 2 requests its per-pixel mask / target / image values together with its staging loads (round 6); 0 = after its
 * convolutions, as rounds 2-5 did.  Process-wide; also GSICP_LOSS_HOIST=0 at load.  Returns the previous value.  Same bits either way
 * (tests/test_mapper_ops_gpu.py). */
int gsicp_mapper_loss_set_hoist(int hoist);

/* gsicp_mapper_loss_indirect that also ADVANCES the optimiser's device step counter (ABI 5): the one thread that finishes the loss value does
 * `if (guard_count && *guard_count > guard_limit) ++*skipped_dev; else ++*step_dev;` — after the forward of the captured iteration (whose
 * duplicate count is the guard) and before its Adam launch, which is then called with bump_step = 2 (gsicp_adam_step_sparse: "already bumped",
 * torch's own order: step += 1, then the update [REF mp_Mapper.py:247 optimizer.step()]) and needs no one-thread bump launch behind it.
 * guard_count / skipped_dev may be NULL. */
int gsicp_mapper_loss_indirect_bump(const float* image, const float* depth, const float* const* gt_slots, int width, int height, float lambda_dssim,
                                    float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out, float* dL_dimage, float* dL_ddepth,
                                    char* scratch, int* step_dev, const unsigned int* guard_count, unsigned int guard_limit, unsigned int* skipped_dev,
                                    void* stream);

/* GaussianModel's activation getters in one launch each way [REF scene/gaussian_model.py:44-56, 105-125], reached from
 * render_3 at [REF gaussian_renderer/__init__.py:263, 273-274]: opacity = sigmoid(opacity_raw) (P), scaling =
 * exp(scaling_raw) (P,3), rotation = rotation_raw / max(||rotation_raw||, 1e-12) (P,4).  All DEVICE float arrays. */
int gsicp_mapper_activations_forward(int P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                                     float* scaling, float* rotation, const int* live_rows_dev, void* stream);
/* Chain rule of the above.  `opacity` / `scaling` are the forward OUTPUTS, rotation_raw the forward input; any dL_d*
 * input may be NULL (treated as zero) and any dL_d*_raw output may be NULL (skipped). */
int gsicp_mapper_activations_backward(int P, const float* opacity, const float* scaling, const float* rotation_raw, const float* dL_dopacity,
                                      const float* dL_dscaling, const float* dL_drotation, float* dL_dopacity_raw, float* dL_dscaling_raw,
                                      float* dL_drotation_raw, const int* live_rows_dev, void* stream);

/* One torch.optim.Adam step (amsgrad off, weight decay 0) over up to 8 tensors in one launch — the six parameter
 * groups of GaussianModel [REF scene/gaussian_model.py:222-231], stepped at [REF mp_Mapper.py:247].  Arrays are HOST
 * arrays of n_groups entries holding DEVICE pointers / element counts / learning rates; `step` is the 1-based step count
 * after the increment (bias corrections 1 - beta^step). */
int gsicp_adam_step(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const long long* numel, const float* lr, float beta1, float beta2, float eps, int step, void* stream);

/* The same step with the step count and the learning rates in DEVICE memory (the counterpart of
 * torch.optim.Adam(capturable=True)): `lr_dev` is a DEVICE double array of n_groups learning rates (double, so that lr / (1 - beta1^t) rounds exactly as on the host path), `step_dev` a DEVICE
 * int holding the number of steps taken so far; the call applies step *step_dev + 1 and then increments it.  No host
 * value other than the pointers is baked into the launches, so a captured HIP graph replays correctly. */
int gsicp_adam_step_capturable(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                               float eps, int* step_dev, void* stream);

/* gsicp_adam_step_capturable with two additions for a captured mapper iteration:
 *  - bump_step: 0 = leave *step_dev alone (several launches share one counter: bump it with the LAST launch only, so that every
 *    bucket of one optimiser step sees the same step number);
 *  - guard: when guard_count (DEVICE) is non-NULL and *guard_count > guard_limit, nothing is updated and the step is not counted;
 *    *skipped_dev (DEVICE, optional, sticky) is incremented instead.  Pass the rasteriser's num_rendered_dev and its list capacity:
 *    a sync-free forward whose duplicate lists overflowed rendered nothing, and its all-zero gradients must not move the parameters on
 *    stale momentum;
 *  - live rows: when live_rows_dev (DEVICE int) is non-NULL, tensor k is updated only in its first *live_rows_dev * row_width[k]
 *    elements (row_width: HOST array of n_groups ints) — the parameters of a capacity-backed map whose live count changes on the device. */
int gsicp_adam_step_guarded(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                            float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                            unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, void* stream);

/* gsicp_adam_step_guarded with a ROW FREEZE: when row_freeze_dev (DEVICE int[rows]) is non-NULL, tensor k with group_frozen[k] != 0 (HOST array of
 * n_groups ints) leaves the rows whose freeze word is non-zero untouched — parameter and both moments, as if those rows were not parameters
 * (row_width is then required).  What the fused in-system mapper uses to keep the geometry of the Gaussians the tracker aligns against
 * (gs_icp_slam_amd/refglue.py, policy `freeze`).  Live-row bound: with a live count the element index is below 2^32. */
int gsicp_adam_step_masked(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                           float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                           unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, const int* row_freeze_dev,
                           const int* group_frozen, void* stream);

/* gsicp_adam_step_masked with SPARSE GRADIENTS (ABI 5): when grad_rows_dev (DEVICE int[rows]; the rasteriser forward's `radii`) is non-NULL, a row
 * with grad_rows_dev[row] <= 0 is a culled Gaussian — its gradient is zero by definition, is NOT read from `grads` (the backward called with
 * raw_params bit 1 does not write it: 26 MB of zero stores and 17 MB of reads per iteration at 300 k Gaussians, 82 % culled) and the update is
 * torch.optim.Adam's on g = 0 (moments decay, the parameter follows its momentum) — the arithmetic the zero-filled rows took (row_width is then
 * required).  bump_step: 0 = leave *step_dev alone, 1 = apply step *step_dev + 1 and advance the counter afterwards, 2 = the counter has ALREADY
 * been advanced in stream order (gsicp_mapper_loss_indirect_bump): apply step *step_dev, no bump. */
int gsicp_adam_step_sparse(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                           float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                           unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, const int* row_freeze_dev,
                           const int* group_frozen, const int* grad_rows_dev, void* stream);

/* Map pruning without reallocation: GaussianModel.prune_points / _prune_optimizer
 * [REF scene/gaussian_model.py:409-447] apply one boolean mask to every parameter, both Adam moments and the per-Gaussian
 * statistics.  This moves the rows with keep[i] != 0 of `n_arrays` (<= 24) row-major DEVICE arrays from src[a] to dst[a] (distinct
 * buffers), preserving order, in three launches for all arrays together.  row_bytes[a] is a multiple of 4; src / dst / row_bytes are
 * HOST arrays of n_arrays entries; `scratch` is a DEVICE buffer of gsicp_store_compact_scratch_bytes(n); *n_out_dev (DEVICE int)
 * receives the number of surviving rows. */
size_t gsicp_store_compact_scratch_bytes(int n);
int gsicp_store_compact(int n, const unsigned char* keep, int n_arrays, const void* const* src, void* const* dst, const int* row_bytes,
                        void* scratch, int* n_out_dev, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * 4b. Multi-GPU mapper (not in the reference, which is single-GPU).  Rank r rasterises the tiles of the 2x2 super-tiles S
 *     with S % tile_mod == r (section 2's tile_mod / tile_rem); these four calls move what the two collectives carry, with static
 *     sizes so that the whole iteration — RCCL calls included — can be captured in a hipGraph.  All pointers are DEVICE pointers.
 *
 *  image all-gather: gsicp_tiles_pack writes this rank's tiles as one chunk of gsicp_tiles_chunk_floats() floats, laid out
 *     [slot k][r, g, b, depth][256 pixels of the tile], slot k = tile (k & 3) of the rank's (k >> 2)-th super-tile (slots past the image edge
 *     are zero padding: 4 * ceil(n_super_tiles / tile_mod) slots); after an all-gather of the tile_mod chunks (rank order),
 *     gsicp_tiles_unpack writes every tile of the (3,H,W) colour and (1,H,W) depth images from the chunk of its owner.
 *  gradient all-reduce: radii (int[P], identical on every rank because every rank preprocesses all Gaussians) selects the rows
 *     with radii > 0; gsicp_rows_pack copies those rows of n_arrays (<= 8) row-major (P, row_width[a]) float arrays, in ascending
 *     Gaussian order, into packed[row_capacity][sum of widths] and sets the flag word packed[row_capacity * sum] to 1.0 when this
 *     rank overflowed (more visible rows than row_capacity, or *guard_count > guard_limit — the rasteriser's duplicate count and
 *     capacity), else 0.0.  Rows behind the visible count are zeroed.  All-reduce (sum) the row_capacity * sum + 1 floats; gsicp_rows_unpack writes the summed rows back in
 *     place (rows with radii <= 0 are left alone: their gradient is exactly zero on every rank) and *overflow_out = 1 when ANY rank
 *     flagged an overflow (pass it to gsicp_adam_step_guarded with guard_limit 0 so every rank skips the same step), else 0.
 *     `scratch` (gsicp_rows_pack_scratch_bytes(P)) carries the scan of pack to unpack.  src / dst / row_width are HOST arrays.
 * ------------------------------------------------------------------------------------------------------ */
size_t gsicp_tiles_chunk_floats(int width, int height, int tile_mod);
int gsicp_tiles_pack(int width, int height, int tile_mod, int tile_rem, const float* color, const float* depth, float* chunk, void* stream);
int gsicp_tiles_unpack(int width, int height, int tile_mod, const float* gathered, float* color, float* depth, void* stream);
size_t gsicp_rows_pack_scratch_bytes(int P);
int gsicp_rows_pack(int P, const int* radii, int n_arrays, const float* const* src, const int* row_width, float* packed, int row_capacity,
                    const unsigned int* guard_count, unsigned int guard_limit, void* scratch, void* stream);
int gsicp_rows_unpack(int P, const int* radii, int n_arrays, float* const* dst, const int* row_width, const float* packed, int row_capacity,
                      const void* scratch, unsigned int* overflow_out, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * 5. Tracker front-end — Tracker.downsample_and_make_pointcloud2 [REF mp_Tracker.py:415-431] in one
 *    launch.  All pointers are DEVICE pointers.  pick_idx (n_pick int64), x_pre, y_pre (n_pick f32) are the arrays
 *    set_downsample_filter builds once [REF mp_Tracker.py:393-413]; depth is the (H*W) raw depth image, depth_type 0 = uint16,
 *    1 = float32; rgb the (H*W*3) uint8 image (may be NULL together with colors).
 *    Outputs (capacity n_pick each): points (n,3) = (x_pre z, y_pre z, z) of the picks with z = depth / depth_scale != 0, in pick
 *    order; colors (n,3) = rgb / 255; z_values (n); trackable_idx (m) = indices INTO those arrays with z <= depth_trunc, ascending;
 *    counts[0] = n, counts[1] = m (device ints).  Feed points / trackable_idx to gsicp_gicp_set_input_source_device and
 *    gsicp_gicp_set_source_track_device.
 * ------------------------------------------------------------------------------------------------------ */
int gsicp_frontend_make_pointcloud(int n_pick, const long long* pick_idx, const float* x_pre, const float* y_pre, const void* depth,
                                   int depth_type, const unsigned char* rgb, float depth_scale, float depth_trunc, float* points,
                                   float* colors, float* z_values, int* trackable_idx, int* counts, void* stream);
/* set_source_filter for a device-resident list: trackable_idx[r] is the index of the r-th trackable source point (what
 * `input_filter[trackable_filter] = 1..m` encodes on the numpy API [REF mp_Tracker.py:192-195]). */
int gsicp_gicp_set_source_track_device(gsicp_gicp*, const int* trackable_idx, int n_track, void* producer_stream, int wait);

#ifdef __cplusplus
}
#endif
#endif /* GSICP_HIP_H */
