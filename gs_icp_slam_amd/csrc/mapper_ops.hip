// Mapper-side operators next to the rasteriser on the reference's per-iteration path:
//   * the mapping loss  0.8 L1 + 0.2 (1 - SSIM) + 0.1 L1(depth / 10)  [REF mp_Mapper.py:225-240] with the reference's
//     masked L1 and 11x11 Gaussian-window SSIM [REF utils/loss_utils.py:17-20, 27-69], forward AND gradient, in two
//     launches instead of the ~60 torch launches (5 depth-wise conv2d + ~25 element-wise ops, and their backward);
//   * one multi-tensor Adam step over the six parameter groups of GaussianModel
//     [REF scene/gaussian_model.py:222-231; mp_Mapper.py:247] in ONE launch.
//
// Loss kernels: one 256-thread workgroup per 16x16 tile and channel.  The 26x26 halo tile goes through LDS once; the
// reference's 11x11 Gaussian window is applied separably (11 + 11 taps, with 1-D weights bit-identical to the reference's) to
// five moment maps (x, y, xx, yy, xy) in pass 1 and to the three derivative maps in pass 2.  Everything is HBM-streaming: pass 1 reads 2 and writes 3 floats per
// pixel-channel, pass 2 reads 5 and writes 1.  Sums are reduced per workgroup and finished in a fixed order (no float atomics:
// results are bit-reproducible) by one workgroup of pass 2 — or by a one-workgroup kernel when only the loss value is asked for.
#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <cstdint>
#include <cstdlib>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/gsicp_hip.h"
#include "raster_common.hpp"

namespace gsicp {
extern thread_local std::string g_last_error;

namespace {

constexpr int LT = 32;            // output tile (32 x 32 pixels per 256-thread workgroup, 4 pixels per thread)
constexpr int HALO = 5;           // 11x11 window
constexpr int LW = LT + 2 * HALO; // 42
constexpr int LWS = 44;           // staged row stride (floats): 16-byte aligned rows for ds_read_b128
constexpr int LHS = 36;           // row stride of the horizontally filtered maps (16-byte aligned float4 stores)
constexpr int HSEG = LT / 4;      // 4-column segments per row in the horizontal pass

struct Win { float w[11]; };   // the reference's normalised 11-tap Gaussian (its 2-D window is the float32 outer product of this)

__device__ inline float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// 16 consecutive staged values -> registers (four 128-bit LDS reads)
__device__ inline void lds_load16(const float* __restrict__ row, float v[16]) {
    const float4* p = (const float4*)row;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = p[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}
// four adjacent outputs of the 11-tap filter from 14 inputs held in registers (same tap order as a scalar loop: k = 0..10).
// Fused multiply-adds written out (one rounding per tap; the compiler's default contraction forms them here as well).
__device__ inline float4 conv4(const float v[16], const Win& win) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float w = win.w[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(w, v[j + k], o[j]);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// Pass 1.  grid = (tiles_x, tiles_y, 4): z = 0..2 colour channels, z = 3 the depth L1 term (no SSIM).
// partial[(z * n_tiles + tile) * 2 + {0,1}] = {sum of masked |diff|, sum of SSIM map} of this workgroup.
// Register tiling: the horizontal pass makes 4 adjacent outputs from 14 staged values (4 x ds_read_b128 per array), the vertical
// pass 4 stacked outputs from 14 rows — a quarter of the LDS traffic of one-output-per-thread, and a 1.7x halo instead of 2.6x.
__global__ __launch_bounds__(256) void loss_pass1_kernel(const float* __restrict__ image, const float* __restrict__ depth,
                                                         const float* __restrict__ gt_image, const float* __restrict__ gt_depth, int W, int H,
                                                         Win win, float d_max, float dS_scale /* = -lambda / (3HW) */,
                                                         float* __restrict__ abc /* (3, 3, H, W): A, B, C per channel */,
                                                         float* __restrict__ partial, const float* const* __restrict__ gt_slots) {
    if (gt_slots) { gt_image = gt_slots[0]; gt_depth = gt_slots[1]; }   // keyframe selected on the device (gsicp_mapper_select_view): two scalar loads
    // LDS overlay: the horizontal pass runs in two sub-phases — the three maps that need x (mu_x, E[xx], E[xy]) first; then x is dead and
    // mu_y is written over it — so four map buffers instead of five: 39 KB instead of 45 KB per workgroup, four workgroups per CU instead
    // of three (the kernel is bound by how many workgroups a CU holds).  Same arithmetic in the same order: bit-identical results.
    __shared__ __attribute__((aligned(16))) float s_x[LW][LWS];
    __shared__ __attribute__((aligned(16))) float s_y[LW][LWS];
    __shared__ __attribute__((aligned(16))) float s_hb[4][LW][LHS];
    static_assert(LW * LHS <= LW * LWS, "a filtered map must fit in a staged map's space");
    // moment order of the vertical pass: 0 mu_x, 1 mu_y, 2 E[xx], 3 E[yy], 4 E[xy]
    float (*const s_h[5])[LHS] = {s_hb[0], (float (*)[LHS])&s_x[0][0], s_hb[1], s_hb[3], s_hb[2]};
    __shared__ float s_red[4][2];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t HW = (size_t)W * H;
    const int n_tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    const int lx = tid & 31, ry = tid >> 5;      // this thread's output column and its group of four rows
    float l1 = 0.f, ssum = 0.f;
    // Round 5 experiment (grid z = 3, GSICP_LOSS_DEPTH_IN_CH0=1, default OFF): the depth term of a tile computed by the tile's channel-0 workgroup after
    // its colour work, by the same threads in the same order (same partial-sum slot, same bits).  The idea: the z = 3 slice is a quarter of the launch's
    // workgroups, each holding a 39 KB LDS allocation for a few loads and a reduction.  Measured: no gain (they leave their slots at once).
    const bool depth_here = ch == 3 || (gridDim.z == 3 && ch == 0);
    float l1d = 0.f;
    if (depth_here) {   // depth term: L1(depth / d_max, gt_depth / d_max), masked where gt == 0 (loads issued before the colour work)
        const int px = x0 + lx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            if (px < W && py < H) {
                const float g = gt_depth[(size_t)py * W + px] / d_max;
                const float d = depth[(size_t)py * W + px] / d_max;
                if (g != 0.f) l1d += fabsf(d - g);
            }
        }
    }
    if (ch == 3) {
        l1 = l1d;
    } else {
        // stage the halo tile: y = gt * (gt_depth > 0), x = where(y != 0, image, 0); zero outside the image.  All loads of the
        // (unrolled) loop are unconditional on clamped addresses and issued before the first use: this phase is pure latency.
        constexpr int NST = (LW * LWS + 255) / 256;
        float gd[NST], gi[NST], im[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + u * 256, r = i / LWS, c = i % LWS;
            int px = x0 + c - HALO, py = y0 + r - HALO;
            px = px < 0 ? 0 : (px >= W ? W - 1 : px);
            py = py < 0 ? 0 : (py >= H ? H - 1 : py);
            const size_t pix = (size_t)py * W + px;
            gd[u] = gt_depth[pix]; gi[u] = gt_image[ch * HW + pix]; im[u] = image[ch * HW + pix];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + u * 256, r = i / LWS, c = i % LWS;
            const int px = x0 + c - HALO, py = y0 + r - HALO;
            if (i < LW * LWS) {
                const bool in = c < LW && px >= 0 && px < W && py >= 0 && py < H;
                const float yv = (in && gd[u] > 0.f) ? gi[u] : 0.f;
                s_x[r][c] = yv != 0.f ? im[u] : 0.f;
                s_y[r][c] = yv;
            }
        }
        __syncthreads();
        // Separable evaluation (11 + 11 taps instead of 121).  The reference convolves with the float32 outer product g g^T; the
        // separable form differs from it only by the rounding of the 121 products (relative 6e-8, zero-mean) PROVIDED the 1-D
        // weights are bit-identical to the reference's (an early version normalised them with a sequentially rounded sum, one ulp
        // off, and that 6e-8 scale error alone moved the SSIM mean by 1e-5 through sigma = E[xx] - mu^2).
        // the L1 term needs this thread's own x, y after x has been overwritten: fetch them now
        float xc[4], yc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xc[j] = s_x[4 * ry + j + HALO][lx + HALO]; yc[j] = s_y[4 * ry + j + HALO][lx + HALO]; }
        for (int it = tid; it < LW * HSEG; it += 256) {
            const int r = it / HSEG, c0 = 4 * (it % HSEG);
            float xv[16], yv[16], pr[16];
            lds_load16(&s_x[r][c0], xv);
            lds_load16(&s_y[r][c0], yv);
            *(float4*)&s_h[0][r][c0] = conv4(xv, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = xv[i] * xv[i];
            *(float4*)&s_h[2][r][c0] = conv4(pr, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = xv[i] * yv[i];
            *(float4*)&s_h[4][r][c0] = conv4(pr, win);
        }
        __syncthreads();   // x is consumed: mu_y may land on it
        for (int it = tid; it < LW * HSEG; it += 256) {
            const int r = it / HSEG, c0 = 4 * (it % HSEG);
            float yv[16], pr[16];
            lds_load16(&s_y[r][c0], yv);
            *(float4*)&s_h[1][r][c0] = conv4(yv, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = yv[i] * yv[i];
            *(float4*)&s_h[3][r][c0] = conv4(pr, win);
        }
        __syncthreads();
        float mom[5][4];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 14; ++i) v[i] = s_h[m][4 * ry + i][lx];
            const float4 o = conv4(v, win);
            mom[m][0] = o.x; mom[m][1] = o.y; mom[m][2] = o.z; mom[m][3] = o.w;
        }
        const int px = x0 + lx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            if (px < W && py < H) {
                const float mu1 = mom[0][j], mu2 = mom[1][j], e11 = mom[2][j], e22 = mom[3][j], e12 = mom[4][j];
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float a = 2.f * mu12 + C1, b = 2.f * (e12 - mu12) + C2;
                const float c = mu1_sq + mu2_sq + C1, d = (e11 - mu1_sq) + (e22 - mu2_sq) + C2;
                // v_rcp_f32 (1 ulp) instead of IEEE divisions (~10 instructions each): c, d >= C1, C2 > 0, far from the denormal range
                const float inv_d = __builtin_amdgcn_rcpf(d);
                const float inv_cd = __builtin_amdgcn_rcpf(c) * inv_d;
                const float S = a * b * inv_cd;
                ssum += S;
                // dS/d(mu1), dS/d(E[xx]), dS/d(E[xy]) with mu2, E[yy] fixed (the target image carries no gradient)
                const float dS_dmu1 = 2.f * mu2 * (b - a) * inv_cd - 2.f * mu1 * S * (d - c) * inv_cd;
                const float dS_de11 = -S * inv_d;
                const float dS_de12 = 2.f * a * inv_cd;
                const size_t pix = (size_t)py * W + px;
                abc[(ch * 3 + 0) * HW + pix] = dS_scale * dS_dmu1;
                abc[(ch * 3 + 1) * HW + pix] = dS_scale * dS_de11;
                abc[(ch * 3 + 2) * HW + pix] = dS_scale * dS_de12;
                if (yc[j] != 0.f) l1 += fabsf(xc[j] - yc[j]);
            }
        }
    }
    l1 = wave_sum_f(l1); ssum = wave_sum_f(ssum);
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = l1; s_red[tid >> 6][1] = ssum; }
    __syncthreads();
    if (tid == 0) {
        partial[((size_t)ch * n_tiles + tile) * 2 + 0] = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        partial[((size_t)ch * n_tiles + tile) * 2 + 1] = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
    }
    if (depth_here && ch != 3) {   // the tile's depth sum into the slot the z = 3 workgroup used to fill (its SSIM word is zero there as well)
        __syncthreads();
        l1d = wave_sum_f(l1d);
        if ((tid & 63) == 0) { s_red[tid >> 6][0] = l1d; s_red[tid >> 6][1] = 0.f; }
        __syncthreads();
        if (tid == 0) {
            partial[((size_t)3 * n_tiles + tile) * 2 + 0] = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
            partial[((size_t)3 * n_tiles + tile) * 2 + 1] = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
        }
    }
}

// Finish the sums in a fixed order and form the loss.  out = {loss, L1, SSIM mean, depth L1}.  256 threads; called either by the
// one-workgroup kernel below (loss value only) or by ONE workgroup of pass 2 (value + gradients: the reduction then costs no launch of its
// own — 4.6 us of a 0.36 ms iteration) — same thread count and order, so both give the same bits.
struct LossReduceArgs { const float2* partial; int n_tiles; float inv_n_img, inv_n_depth, lambda_dssim, depth_weight; float* out;
                        float lambda_const;      // the loss's constant term lambda * 1: lambda on one GPU; lambda / N per rank when N ranks each sum their own blocks
                        // STEP BUMP (round 6; gsicp_mapper_loss_indirect_bump): the optimiser's device step counter is advanced HERE, by the one thread
                        // that finishes the loss value — after the forward (whose duplicate count is the overflow guard) and before the Adam launch
                        // of the same captured iteration, which then reads the already advanced count (torch's own order: step += 1, then the update)
                        // and needs no one-thread bump kernel behind it (4.1 us of launch floor per iteration).  Under a tripped guard the step is
                        // void: the skipped-steps counter advances instead.
                        int* bump_step; const unsigned* bump_guard; unsigned bump_limit; unsigned* bump_skipped; };
__device__ inline void loss_reduce_body(const LossReduceArgs& q, const int tid) {
    __shared__ double s_acc[3][4];
    const int lane = tid & 63, wave = tid >> 6;
    double l1 = 0, ss = 0, ld = 0;
    const int n_img = 3 * q.n_tiles, n_all = 4 * q.n_tiles;
    for (int base = 0; base < n_all; base += 4 * 256) {   // four independent loads in flight per thread (the loop is latency-bound)
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * 256 + tid;
            v[u] = i < n_all ? q.partial[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * 256 + tid;
            if (i < n_img) { l1 += (double)v[u].x; ss += (double)v[u].y; }
            else if (i < n_all) ld += (double)v[u].x;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_down(l1, off, 64); ss += __shfl_down(ss, off, 64); ld += __shfl_down(ld, off, 64); }
    if (lane == 0) { s_acc[0][wave] = l1; s_acc[1][wave] = ss; s_acc[2][wave] = ld; }
    __syncthreads();
    if (tid == 0) {
        double a = 0, b = 0, c = 0;
        for (int w = 0; w < 4; ++w) { a += s_acc[0][w]; b += s_acc[1][w]; c += s_acc[2][w]; }
        const float L1 = (float)(a * q.inv_n_img), SS = (float)(b * q.inv_n_img), LD = (float)(c * q.inv_n_depth);
        q.out[0] = q.lambda_const == q.lambda_dssim ? (1.f - q.lambda_dssim) * L1 + q.lambda_dssim * (1.f - SS) + q.depth_weight * LD
                                                    : ((1.f - q.lambda_dssim) * L1 + (q.lambda_const - q.lambda_dssim * SS)) + q.depth_weight * LD;
        q.out[1] = L1; q.out[2] = SS; q.out[3] = LD;
        if (q.bump_step) {
            if (q.bump_guard && *q.bump_guard > q.bump_limit) { if (q.bump_skipped) *q.bump_skipped += 1u; }
            else *q.bump_step += 1;
        }
    }
}
__global__ __launch_bounds__(256) void loss_reduce_kernel(LossReduceArgs q) { loss_reduce_body(q, (int)threadIdx.x); }

// Pass 2.  dL/dx(p) = sum_q w(q - p) [A_q + 2 x_p B_q + y_p C_q]  (+ the L1 sign term), masked where y == 0.  Same tiling as pass 1.
template <bool HOIST>
__global__ __launch_bounds__(256) void loss_pass2_kernel(const float* __restrict__ image, const float* __restrict__ depth,
                                                         const float* __restrict__ gt_image, const float* __restrict__ gt_depth, int W, int H,
                                                         Win win, float d_max, float l1_scale /* (1 - lambda) / (3HW) */,
                                                         float depth_scale /* w_d / (HW d_max) */, const float* __restrict__ abc,
                                                         float* __restrict__ dL_dimage, float* __restrict__ dL_ddepth, LossReduceArgs red,
                                                         const float* const* __restrict__ gt_slots) {
    if (gt_slots) { gt_image = gt_slots[0]; gt_depth = gt_slots[1]; }
    // LDS overlay: the horizontally filtered map m is written where the INPUT map m - 1 lay (already consumed; map 0 gets a buffer of its
    // own) — 28 KB instead of 40 KB per workgroup, five workgroups per CU instead of four (these kernels are bound by how many
    // workgroups a CU can hold, not by bandwidth or arithmetic), at the price of a barrier per map instead of one for all three.
    __shared__ __attribute__((aligned(16))) float s_in[3][LW][LWS];
    __shared__ __attribute__((aligned(16))) float s_h0[LW][LHS];
    static_assert(LW * LHS <= LW * LWS, "a filtered map must fit in an input map's space");
    float (*const s_hm[3])[LHS] = {s_h0, (float (*)[LHS])&s_in[0][0][0], (float (*)[LHS])&s_in[1][0][0]};
    const int tid = threadIdx.x, lx = tid & 31, ry = tid >> 5;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t HW = (size_t)W * H;
    const int px = x0 + lx;
    const bool depth_here = ch == 3 || (gridDim.z == 3 && ch == 0);     // round 5: grid z = 3, the depth gradient of a tile by its channel-0 workgroup
    if (depth_here) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            if (px < W && py < H) {
                const size_t pix = (size_t)py * W + px;
                const float g = gt_depth[pix] / d_max, d = depth[pix] / d_max;
                float gr = 0.f;
                if (g != 0.f) gr = d > g ? depth_scale : (d < g ? -depth_scale : 0.f);
                dL_ddepth[pix] = gr;
            }
        }
    }
    if (ch == 3) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && red.out) loss_reduce_body(red, tid);   // pass 1's partial sums are complete: finish the loss value here
        return;
    }
    // HOIST (round 6, default; GSICP_LOSS_HOIST=0 for the A/B partner): this thread's mask / target / image values, needed after the convolutions, are requested HERE, with the
    // staging loads — the kernel otherwise ends every workgroup with an exposed round trip to memory
    float h_gd[4] = {0.f, 0.f, 0.f, 0.f}, h_gt[4] = {0.f, 0.f, 0.f, 0.f}, h_im[4] = {0.f, 0.f, 0.f, 0.f};
    if (HOIST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            const size_t pix = (px < W && py < H) ? (size_t)py * W + px : 0;
            h_gd[j] = gt_depth[pix]; h_gt[j] = gt_image[ch * HW + pix]; h_im[j] = image[ch * HW + pix];
        }
    }
    {
        constexpr int NST = (LW * LWS + 255) / 256;
        float va[NST], vb[NST], vc[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {   // unconditional loads on clamped addresses, all in flight together
            const int i = tid + u * 256, r = i / LWS, c = i % LWS;
            int qx = x0 + c - HALO, qy = y0 + r - HALO;
            qx = qx < 0 ? 0 : (qx >= W ? W - 1 : qx);
            qy = qy < 0 ? 0 : (qy >= H ? H - 1 : qy);
            const size_t q = (size_t)qy * W + qx;
            va[u] = abc[(ch * 3 + 0) * HW + q]; vb[u] = abc[(ch * 3 + 1) * HW + q]; vc[u] = abc[(ch * 3 + 2) * HW + q];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + u * 256, r = i / LWS, c = i % LWS;
            const int qx = x0 + c - HALO, qy = y0 + r - HALO;
            if (i < LW * LWS) {
                const bool in = c < LW && qx >= 0 && qx < W && qy >= 0 && qy < H;
                s_in[0][r][c] = in ? va[u] : 0.f; s_in[1][r][c] = in ? vb[u] : 0.f; s_in[2][r][c] = in ? vc[u] : 0.f;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        for (int it = tid; it < LW * HSEG; it += 256) {
            const int r = it / HSEG, c0 = 4 * (it % HSEG);
            float v[16];
            lds_load16(&s_in[m][r][c0], v);
            *(float4*)&s_hm[m][r][c0] = conv4(v, win);
        }
        __syncthreads();   // input map m is consumed: the next map's output may land on it
    }
    float acc[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m) {   // symmetric window: correlation == convolution
        float v[16];
#pragma unroll
        for (int i = 0; i < 14; ++i) v[i] = s_hm[m][4 * ry + i][lx];
        const float4 o = conv4(v, win);
        acc[m][0] = o.x; acc[m][1] = o.y; acc[m][2] = o.z; acc[m][3] = o.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int py = y0 + 4 * ry + j;
        if (px < W && py < H) {
            const size_t pix = (size_t)py * W + px;
            const float yv = HOIST ? (h_gd[j] > 0.f ? h_gt[j] : 0.f) : (gt_depth[pix] > 0.f ? gt_image[ch * HW + pix] : 0.f);
            float gr = 0.f;
            if (yv != 0.f) {
                const float xv = HOIST ? h_im[j] : image[ch * HW + pix];
                gr = acc[0][j] + 2.f * xv * acc[1][j] + yv * acc[2][j];
                gr += xv > yv ? l1_scale : (xv < yv ? -l1_scale : 0.f);
            }
            dL_dimage[ch * HW + pix] = gr;
        }
    }
    if (gridDim.z == 3 && ch == 0 && blockIdx.x == 0 && blockIdx.y == 0 && red.out) {
        __syncthreads();
        loss_reduce_body(red, tid);   // pass 1's partial sums are complete: finish the loss value here (same 256 threads, same order: same bits)
    }
}

// ------------------------------------------------------------------------------------------------ fused loss (value parts + gradient)
// ONE kernel instead of pass 1 + pass 2: a workgroup stages its 32x32 tile with a TEN-pixel halo (52x52), builds the five window moments on
// the 42x42 region its outputs' windows reach, forms the SSIM map and the three derivative maps there, and applies the window to them for
// its own 32x32 pixels — the derivative maps never travel through memory.  Against the two-pass form: no (3,3,H,W) map buffer (36 B / pixel
// written, 65 B / pixel re-read with the halo), no second read of the images; the price is the window moments on 1.7x the pixels.
// Same per-pixel arithmetic in the same order as the two passes: gradients are bit-identical to theirs.
constexpr int FH = 2 * HALO;           // staged halo
constexpr int FIN = LT + 2 * FH;       // 52 staged rows / columns
constexpr int FMID = LT + 2 * HALO;    // 42 rows / columns of moments and derivative maps
constexpr int FXS = 56;                // staged row stride (floats)
constexpr int FMS = 44;                // row stride of the horizontally filtered moment maps and of the derivative maps
constexpr int FSEG = FMS / 4;          // 11 four-column segments across the region
constexpr int FVG = (FMID + 3) / 4;    // 11 four-row groups down the region

__global__ __launch_bounds__(256) void loss_fused_kernel(const float* __restrict__ image, const float* __restrict__ depth,
                                                         const float* __restrict__ gt_image, const float* __restrict__ gt_depth, int W, int H,
                                                         Win win, float d_max, float dS_scale /* -lambda / (3HW) */, float l1_scale /* (1 - lambda) / (3HW) */,
                                                         float depth_scale /* w_d / (HW d_max) */, float* __restrict__ dL_dimage,
                                                         float* __restrict__ dL_ddepth, float* __restrict__ partial, int tile_mod, int tile_rem,
                                                         const float* const* __restrict__ gt_slots) {
    if (gt_slots) { gt_image = gt_slots[0]; gt_depth = gt_slots[1]; }
    // x and y staging as ONE object: the three derivative maps (FMID x FMS each) later overlay BOTH, and only members of one array are
    // guaranteed contiguous (separate __shared__ variables may be laid out in any order — ADVICE r3)
    __shared__ __attribute__((aligned(16))) float s_xy[2][FIN][FXS];
    float (*const s_x)[FXS] = s_xy[0];
    float (*const s_y)[FXS] = s_xy[1];
    __shared__ __attribute__((aligned(16))) float s_b[3][FIN][FMS];     // three horizontally filtered moment maps at a time; later the filtered derivative maps
    __shared__ float s_red[4][2];
    static_assert(3 * FMID * FMS <= 2 * FIN * FXS, "derivative maps must fit in the staging buffers");
    static_assert(3 * FMID * LHS <= 3 * FIN * FMS, "filtered derivative maps must fit in the moment buffers");
    float (*const s_a)[FMID][FMS] = (float (*)[FMID][FMS])&s_xy[0][0][0];
    float (*const s_h2)[FMID][LHS] = (float (*)[FMID][LHS])&s_b[0][0][0];
    const int tid = threadIdx.x, lx = tid & 31, ry = tid >> 5;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT, ch = blockIdx.z;
    const size_t HW = (size_t)W * H;
    const int n_tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    const int px = x0 + lx;
    float l1 = 0.f, ssum = 0.f;
    // multi-GPU: a 32x32 block is one 2x2 super-tile of the rasteriser (raster_common.hpp tile_xy_is_mine); only its owner works on it, the
    // others contribute zero partial sums and leave its pixels of dL_dimage / dL_ddepth alone (nobody reads them on this rank)
    if (tile_mod > 1 && !(tile_mod_is_band(tile_mod) ? super_row_is_mine((int)blockIdx.y, tile_mod) : (tile % tile_mod) == tile_rem)) {
        if (tid == 0) { partial[((size_t)ch * n_tiles + tile) * 2 + 0] = 0.f; partial[((size_t)ch * n_tiles + tile) * 2 + 1] = 0.f; }
        return;
    }
    if (ch == 3) {   // depth term: value and gradient
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            if (px < W && py < H) {
                const size_t pix = (size_t)py * W + px;
                const float g = gt_depth[pix] / d_max, d = depth[pix] / d_max;
                float gr = 0.f;
                if (g != 0.f) { l1 += fabsf(d - g); gr = d > g ? depth_scale : (d < g ? -depth_scale : 0.f); }
                if (dL_ddepth) dL_ddepth[pix] = gr;
            }
        }
    } else {
        {   // stage: y = gt * (gt_depth > 0), x = where(y != 0, image, 0); zero outside the image.  Loads unconditional on clamped addresses.
            constexpr int NST = (FIN * FXS + 255) / 256;   // 12
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float gd[NST / 2], gi[NST / 2], im[NST / 2];
#pragma unroll
                for (int u = 0; u < NST / 2; ++u) {
                    const int i = tid + (half * (NST / 2) + u) * 256, r = i / FXS, c = i % FXS;
                    int qx = x0 + c - FH, qy = y0 + r - FH;
                    qx = qx < 0 ? 0 : (qx >= W ? W - 1 : qx);
                    qy = qy < 0 ? 0 : (qy >= H ? H - 1 : qy);
                    const size_t q = (size_t)qy * W + qx;
                    gd[u] = gt_depth[q]; gi[u] = gt_image[ch * HW + q]; im[u] = image[ch * HW + q];
                }
#pragma unroll
                for (int u = 0; u < NST / 2; ++u) {
                    const int i = tid + (half * (NST / 2) + u) * 256, r = i / FXS, c = i % FXS;
                    const int qx = x0 + c - FH, qy = y0 + r - FH;
                    if (i < FIN * FXS) {
                        const bool in = c < FIN && qx >= 0 && qx < W && qy >= 0 && qy < H;
                        const float yv = (in && gd[u] > 0.f) ? gi[u] : 0.f;
                        s_x[r][c] = yv != 0.f ? im[u] : 0.f;
                        s_y[r][c] = yv;
                    }
                }
            }
        }
        __syncthreads();
        float xc[4], yc[4];     // this thread's own four pixels (the staging buffers are overwritten below)
#pragma unroll
        for (int j = 0; j < 4; ++j) { xc[j] = s_x[4 * ry + j + FH][lx + FH]; yc[j] = s_y[4 * ry + j + FH][lx + FH]; }
        // this thread's vertical-pass items: four stacked rows (4 vg .. 4 vg + 3) of one column of the 42x42 region
        const int itA = tid, itB = tid + 256;
        const bool hasB = itB < FMID * FVG;
        const int colA = itA % FMID, vgA = itA / FMID, colB = hasB ? itB % FMID : 0, vgB = hasB ? itB / FMID : 0;
        float mom[2][5][4];
        auto vpass = [&](const int m_buf, const int m_out) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && !hasB) break;
                const int col = k ? colB : colA, vg = k ? vgB : vgA;
                float v[16];
#pragma unroll
                for (int i = 0; i < 14; ++i) { const int r = 4 * vg + i; v[i] = s_b[m_buf][r < FIN ? r : FIN - 1][col]; }
                const float4 o = conv4(v, win);
                mom[k][m_out][0] = o.x; mom[k][m_out][1] = o.y; mom[k][m_out][2] = o.z; mom[k][m_out][3] = o.w;
            }
        };
        // horizontal pass A: the maps that need x — mu_x, E[xx], E[xy]
        for (int it = tid; it < FIN * FSEG; it += 256) {
            const int r = it / FSEG, c0 = 4 * (it % FSEG);
            float xv[16], yv[16], pr[16];
            lds_load16(&s_x[r][c0], xv);
            lds_load16(&s_y[r][c0], yv);
            *(float4*)&s_b[0][r][c0] = conv4(xv, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = xv[i] * xv[i];
            *(float4*)&s_b[1][r][c0] = conv4(pr, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = xv[i] * yv[i];
            *(float4*)&s_b[2][r][c0] = conv4(pr, win);
        }
        __syncthreads();
        vpass(0, 0); vpass(1, 2); vpass(2, 4);
        __syncthreads();
        // horizontal pass B: mu_y, E[yy]
        for (int it = tid; it < FIN * FSEG; it += 256) {
            const int r = it / FSEG, c0 = 4 * (it % FSEG);
            float yv[16], pr[16];
            lds_load16(&s_y[r][c0], yv);
            *(float4*)&s_b[0][r][c0] = conv4(yv, win);
#pragma unroll
            for (int i = 0; i < 14; ++i) pr[i] = yv[i] * yv[i];
            *(float4*)&s_b[1][r][c0] = conv4(pr, win);
        }
        __syncthreads();
        vpass(0, 1); vpass(1, 3);
        // SSIM map + derivative maps on the 42x42 region (x and y staging is dead: the maps land there)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !hasB) break;
            const int col = k ? colB : colA, vg = k ? vgB : vgA;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * vg + j;
                if (r >= FMID) continue;
                const int qx = x0 + col - HALO, qy = y0 + r - HALO;
                const bool in = qx >= 0 && qx < W && qy >= 0 && qy < H;
                const float mu1 = mom[k][0][j], mu2 = mom[k][1][j], e11 = mom[k][2][j], e22 = mom[k][3][j], e12 = mom[k][4][j];
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float a = 2.f * mu12 + C1, b = 2.f * (e12 - mu12) + C2;
                const float c = mu1_sq + mu2_sq + C1, d = (e11 - mu1_sq) + (e22 - mu2_sq) + C2;
                const float inv_d = __builtin_amdgcn_rcpf(d);
                const float inv_cd = __builtin_amdgcn_rcpf(c) * inv_d;
                const float S = a * b * inv_cd;
                const float dS_dmu1 = 2.f * mu2 * (b - a) * inv_cd - 2.f * mu1 * S * (d - c) * inv_cd;
                const float dS_de11 = -S * inv_d;
                const float dS_de12 = 2.f * a * inv_cd;
                s_a[0][r][col] = in ? dS_scale * dS_dmu1 : 0.f;
                s_a[1][r][col] = in ? dS_scale * dS_de11 : 0.f;
                s_a[2][r][col] = in ? dS_scale * dS_de12 : 0.f;
                // the SSIM mean counts every image pixel once: by the workgroup whose OUTPUT tile holds it
                if (in && r >= HALO && r < HALO + LT && col >= HALO && col < HALO + LT) ssum += S;
            }
        }
        __syncthreads();
        // window applied to the derivative maps: horizontal ...
        for (int it = tid; it < FMID * HSEG; it += 256) {
            const int r = it / HSEG, c0 = 4 * (it % HSEG);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float v[16];
                lds_load16(&s_a[m][r][c0], v);
                *(float4*)&s_h2[m][r][c0] = conv4(v, win);
            }
        }
        __syncthreads();
        // ... and vertical, for this thread's own four pixels
        float acc[3][4];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 14; ++i) v[i] = s_h2[m][4 * ry + i][lx];
            const float4 o = conv4(v, win);
            acc[m][0] = o.x; acc[m][1] = o.y; acc[m][2] = o.z; acc[m][3] = o.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int py = y0 + 4 * ry + j;
            if (px < W && py < H) {
                float gr = 0.f;
                if (yc[j] != 0.f) {
                    l1 += fabsf(xc[j] - yc[j]);
                    gr = acc[0][j] + 2.f * xc[j] * acc[1][j] + yc[j] * acc[2][j];
                    gr += xc[j] > yc[j] ? l1_scale : (xc[j] < yc[j] ? -l1_scale : 0.f);
                }
                if (dL_dimage) dL_dimage[ch * HW + (size_t)py * W + px] = gr;
            }
        }
    }
    l1 = wave_sum_f(l1); ssum = wave_sum_f(ssum);
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = l1; s_red[tid >> 6][1] = ssum; }
    __syncthreads();
    if (tid == 0) {
        partial[((size_t)ch * n_tiles + tile) * 2 + 0] = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        partial[((size_t)ch * n_tiles + tile) * 2 + 1] = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
    }
}

// ------------------------------------------------------------------------------------------------ map compaction (prune)
// GaussianModel.prune_points [REF scene/gaussian_model.py:409-447] filters every parameter, both Adam moments and the per-Gaussian
// statistics with a boolean mask — ~20 boolean-index launches, each allocating its result.  Here: one order-preserving stream
// compaction (count -> single-workgroup scan -> scatter) moves the surviving rows of ALL arrays from one preallocated buffer set
// to the other (gs_icp_slam_amd/gaussian_store.py keeps two sets and swaps them), so nothing is allocated and addresses are stable.
constexpr int COMPACT_MAX_ARRAYS = 24;
struct CompactTable {
    const unsigned* src[COMPACT_MAX_ARRAYS];   // row-major arrays, rows are multiples of 4 bytes
    unsigned* dst[COMPACT_MAX_ARRAYS];
    int row_words[COMPACT_MAX_ARRAYS];
    int n_arrays;
};
__global__ __launch_bounds__(256) void compact_count_kernel(int n, const unsigned char* __restrict__ keep, unsigned* __restrict__ block_count) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(i < n && keep[i] != 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void compact_scan_kernel(int nblocks, unsigned* __restrict__ block_count, int* __restrict__ n_out) {
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
    if (tid == 1023) *n_out = (int)s_part[1023];
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(int n, const unsigned char* __restrict__ keep, const unsigned* __restrict__ block_base,
                                                              CompactTable t) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool k = i < n && keep[i] != 0;
    const unsigned long long b = __ballot(k);
    if (lane == 0) s_w[wave] = (unsigned)__popcll(b);
    __syncthreads();
    if (!k) return;
    unsigned pos = block_base[blockIdx.x] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    for (int a = 0; a < t.n_arrays; ++a) {
        const int rw = t.row_words[a];
        const unsigned* __restrict__ sp = t.src[a] + (size_t)i * rw;
        unsigned* __restrict__ dp = t.dst[a] + (size_t)pos * rw;
        for (int w = 0; w < rw; ++w) dp[w] = sp[w];
    }
}

// ------------------------------------------------------------------------------------------------ keyframe selection
// The captured mapper iteration reads its camera and target images from fixed device buffers; selecting the keyframe of the
// next replay is ONE launch that refreshes all five (five separate runtime copies cost ~5 us each of fixed latency).
__global__ __launch_bounds__(256) void set_view_kernel(const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
                                                       const float4* __restrict__ gt_image, const float4* __restrict__ gt_depth, size_t n4_image,
                                                       size_t n4_depth, float* __restrict__ d_view, float* __restrict__ d_proj,
                                                       float* __restrict__ d_campos, float4* __restrict__ d_gt_image, float4* __restrict__ d_gt_depth) {
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 16) d_view[t] = view[t];
        else if (t < 32) d_proj[t - 16] = proj[t - 16];
        else if (t < 35) d_campos[t - 32] = campos[t - 32];
    }
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4_image; i += stride) d_gt_image[i] = gt_image[i];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4_depth; i += stride) d_gt_depth[i] = gt_depth[i];
}

// Round 5: the keyframe's images stay where they are.  The captured loss kernels read the two ground-truth pointers from a DEVICE slot pair
// (gsicp_mapper_loss_indirect); selecting a keyframe writes 35 floats and two pointers — no 13 MB copy (6.8 us and 20 MB of traffic per iteration).
__global__ __launch_bounds__(64) void select_view_kernel(const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
                                                         const float* gt_image, const float* gt_depth, float* __restrict__ d_view,
                                                         float* __restrict__ d_proj, float* __restrict__ d_campos, const float** __restrict__ d_slots) {
    const int t = threadIdx.x;
    if (t < 16) d_view[t] = view[t];
    else if (t < 32) d_proj[t - 16] = proj[t - 16];
    else if (t < 35) d_campos[t - 32] = campos[t - 32];
    else if (t == 35) d_slots[0] = gt_image;
    else if (t == 36) d_slots[1] = gt_depth;
}

// Round 6: the same launch also clears the per-call counter region of the CAPTURED rasteriser forward that follows it (`zero_words` 4-byte words at `zero`: per-tile counts,
// cursors, the slot / visible counters — 6.5 k words at 1200x680), so that the forward inside the graph needs no zero-fill launch of its own (4.9 us of launch floor per
// iteration; its is_used array is cleared by the preprocess kernel).  One 256-thread workgroup.
__global__ __launch_bounds__(256) void select_view_zero_kernel(const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
                                                              const float* gt_image, const float* gt_depth, float* __restrict__ d_view,
                                                              float* __restrict__ d_proj, float* __restrict__ d_campos, const float** __restrict__ d_slots,
                                                              uint32_t* __restrict__ zero, size_t zero_words) {
    const int t = threadIdx.x;
    if (t < 16) d_view[t] = view[t];
    else if (t < 32) d_proj[t - 16] = proj[t - 16];
    else if (t < 35) d_campos[t - 32] = campos[t - 32];
    else if (t == 35) d_slots[0] = gt_image;
    else if (t == 36) d_slots[1] = gt_depth;
    for (size_t i = (size_t)t; i < zero_words; i += 256) zero[i] = 0u;
}

// ------------------------------------------------------------------------------------------------ Adam
constexpr int ADAM_MAX_GROUPS = 8;
struct AdamTable {
    float* p[ADAM_MAX_GROUPS];
    const float* g[ADAM_MAX_GROUPS];
    float* m[ADAM_MAX_GROUPS];
    float* v[ADAM_MAX_GROUPS];
    long long end[ADAM_MAX_GROUPS];    // exclusive prefix end of each group in the flattened index space
    float step_size[ADAM_MAX_GROUPS];  // lr / (1 - beta1^t)
    int src[ADAM_MAX_GROUPS];          // index of the group in the caller's arrays (empty tensors are squeezed out)
    int row_width[ADAM_MAX_GROUPS];    // elements per map row (live-row mode), 0 = whole tensor
    int frozen[ADAM_MAX_GROUPS];       // 1: rows with row_freeze[row] != 0 of this tensor are left alone (parameter and both moments)
    int step_offset;                   // capturable path: the update uses step = *step_dev + step_offset (1: the bump follows; 0: it has already happened)
    int n;
};

struct AdamPMV { float p, m, v; };
// one element of torch.optim.Adam; both kernels go through this function so that they round identically
__device__ inline AdamPMV adam_update(float p, float m, float v, float g, float beta1, float beta2, float eps, float inv_bc2_sqrt, float step_size) {
    AdamPMV o;
    o.m = m + (1.f - beta1) * (g - m);
    o.v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(o.v) * inv_bc2_sqrt + eps;
    o.p = p - step_size * (o.m / denom);
    return o;
}
// torch.optim.Adam (amsgrad off, weight decay 0, maximize off):  m += (1-b1)(g-m);  v = b2 v + (1-b2) g g;
// p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// grid = (x, tensor): blockIdx.y picks the tensor, so no per-element group lookup, and each lane moves 16 bytes per access (the update
// is pure streaming: 7 float accesses per element, 118 MB at P = 300 k).  Tensors that are not 16-byte aligned take the scalar loop.
// CAPTURABLE: the step count and the learning rates come from device memory (a captured hipGraph replays them); otherwise the host
// has already folded them into t.step_size / inv_bc2_sqrt_host.
// GUARD (capturable path only): when *guard_count > guard_limit the iteration that produced these gradients is void — the sync-free
// rasteriser forward rendered nothing because its duplicate lists outgrew their capacity — and the step must not happen: parameters,
// both moments and the step count stay as they are, and the bump kernel counts the skipped step in a sticky device counter.
template <bool CAPTURABLE>
__device__ __forceinline__ void adam_tensor_body(const AdamTable& t, const double* __restrict__ lr_dev, const int* step_dev, float beta1, float beta2,
                                                 float eps, float inv_bc2_sqrt_host, const int* __restrict__ live_rows, double* s_bc1_p,
                                                 float* s_inv_bc2_sqrt_p, const int* __restrict__ row_freeze, const int* __restrict__ grad_rows) {
    double& s_bc1 = *s_bc1_p;
    float& s_inv_bc2_sqrt = *s_inv_bc2_sqrt_p;
    if (CAPTURABLE) {
        if (threadIdx.x == 0) {   // two double pow() per BLOCK, not per thread
            const int step = *step_dev + t.step_offset;
            s_bc1 = 1.0 - pow((double)beta1, (double)step);
            s_inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
        }
        __syncthreads();
    }
    const int gidx = blockIdx.y;
    if (gidx >= t.n) return;
    const float inv_bc2_sqrt = CAPTURABLE ? s_inv_bc2_sqrt : inv_bc2_sqrt_host;
    const float step_size = CAPTURABLE ? (float)(lr_dev[t.src[gidx]] / s_bc1)   // the host path's formula, in double like torch's
                                       : t.step_size[gidx];
    long long numel = t.end[gidx] - (gidx ? t.end[gidx - 1] : 0);
    if (live_rows && t.row_width[gidx] > 0) {                  // capacity-backed map: only the live rows are parameters
        const long long live = (long long)*live_rows * t.row_width[gidx];
        numel = live < numel ? live : numel;
    }
    float* __restrict__ P = t.p[gidx];
    const float* __restrict__ G = t.g[gidx];
    float* __restrict__ M = t.m[gidx];
    float* __restrict__ V = t.v[gidx];
    const bool aligned = ((((uintptr_t)P) | ((uintptr_t)G) | ((uintptr_t)M) | ((uintptr_t)V)) & 15) == 0;
    const long long nvec = aligned ? numel / 4 : 0;
    const long long stride = (long long)gridDim.x * 256;
    // ROW FREEZE (round 5; refglue's default policy): a tensor flagged in t.frozen skips the rows whose row_freeze word is non-zero — parameter
    // and both moments stay as they are, exactly as if the row were no parameter at all.
    const unsigned rw = (row_freeze && t.frozen[gidx] && t.row_width[gidx] > 0) ? (unsigned)t.row_width[gidx] : 0u;
    // SPARSE GRADIENTS (round 6): `grad_rows` = the forward's radii.  A row with grad_rows[row] <= 0 is a culled Gaussian: its gradient is ZERO by
    // definition and was not written by the backward (PreprocessBwdArgs.sparse_grads) — it is not read here either; the update is torch.optim.Adam's on
    // g = 0 (the moments decay, the parameter follows its momentum), the same arithmetic the zero-filled rows took.
    const unsigned gw = (grad_rows && t.row_width[gidx] > 0) ? (unsigned)t.row_width[gidx] : 0u;
    if (gw) {
        // rows of the four elements of a 16-byte unit: gw = 4 -> row i; gw = 1 -> rows 4i .. 4i + 3; gw = 3 -> (4i + k) / 3 by a multiply-high
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
            const unsigned e = 4u * (unsigned)i;
            unsigned r0, r1, r2, r3;
            if (gw == 4u) { r0 = r1 = r2 = r3 = (unsigned)i; }
            else if (gw == 1u) { r0 = e; r1 = e + 1u; r2 = e + 2u; r3 = e + 3u; }
            else if (gw == 3u) { r0 = __umulhi(e, 0xAAAAAAABu) >> 1; r1 = __umulhi(e + 1u, 0xAAAAAAABu) >> 1; r2 = __umulhi(e + 2u, 0xAAAAAAABu) >> 1; r3 = __umulhi(e + 3u, 0xAAAAAAABu) >> 1; }
            else { r0 = e / gw; r1 = (e + 1u) / gw; r2 = (e + 2u) / gw; r3 = (e + 3u) / gw; }
            const bool l0 = grad_rows[r0] > 0, l3 = grad_rows[r3] > 0;
            const bool l1 = r1 == r0 ? l0 : (r1 == r3 ? l3 : grad_rows[r1] > 0), l2 = r2 == r3 ? l3 : (r2 == r0 ? l0 : grad_rows[r2] > 0);
            float4 p = ((float4*)P)[i], m = ((float4*)M)[i], v = ((float4*)V)[i];
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l0 | l1 | l2 | l3) {
                const float4 gl = ((const float4*)G)[i];
                g = make_float4(l0 ? gl.x : 0.f, l1 ? gl.y : 0.f, l2 ? gl.z : 0.f, l3 ? gl.w : 0.f);
            }
            AdamPMV a0 = adam_update(p.x, m.x, v.x, g.x, beta1, beta2, eps, inv_bc2_sqrt, step_size);
            AdamPMV a1 = adam_update(p.y, m.y, v.y, g.y, beta1, beta2, eps, inv_bc2_sqrt, step_size);
            AdamPMV a2 = adam_update(p.z, m.z, v.z, g.z, beta1, beta2, eps, inv_bc2_sqrt, step_size);
            AdamPMV a3 = adam_update(p.w, m.w, v.w, g.w, beta1, beta2, eps, inv_bc2_sqrt, step_size);
            if (rw) {
                if (row_freeze[r0] != 0) { a0.p = p.x; a0.m = m.x; a0.v = v.x; }
                if (row_freeze[r1] != 0) { a1.p = p.y; a1.m = m.y; a1.v = v.y; }
                if (row_freeze[r2] != 0) { a2.p = p.z; a2.m = m.z; a2.v = v.z; }
                if (row_freeze[r3] != 0) { a3.p = p.w; a3.m = m.w; a3.v = v.w; }
            }
            ((float4*)M)[i] = make_float4(a0.m, a1.m, a2.m, a3.m);
            ((float4*)V)[i] = make_float4(a0.v, a1.v, a2.v, a3.v);
            ((float4*)P)[i] = make_float4(a0.p, a1.p, a2.p, a3.p);
        }
        for (long long j = 4 * nvec + (long long)blockIdx.x * 256 + threadIdx.x; j < numel; j += stride) {
            const unsigned row = (unsigned)j / gw;
            if (rw && row_freeze[row] != 0) continue;
            const AdamPMV a0 = adam_update(P[j], M[j], V[j], grad_rows[row] > 0 ? G[j] : 0.f, beta1, beta2, eps, inv_bc2_sqrt, step_size);
            M[j] = a0.m; V[j] = a0.v; P[j] = a0.p;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        float4 p = ((float4*)P)[i], m = ((float4*)M)[i], v = ((float4*)V)[i];
        const float4 g = ((const float4*)G)[i];
        AdamPMV a0 = adam_update(p.x, m.x, v.x, g.x, beta1, beta2, eps, inv_bc2_sqrt, step_size);
        AdamPMV a1 = adam_update(p.y, m.y, v.y, g.y, beta1, beta2, eps, inv_bc2_sqrt, step_size);
        AdamPMV a2 = adam_update(p.z, m.z, v.z, g.z, beta1, beta2, eps, inv_bc2_sqrt, step_size);
        AdamPMV a3 = adam_update(p.w, m.w, v.w, g.w, beta1, beta2, eps, inv_bc2_sqrt, step_size);
        if (rw) {
            const unsigned e = 4u * (unsigned)i;
            if (row_freeze[e / rw] != 0) { a0.p = p.x; a0.m = m.x; a0.v = v.x; }
            if (row_freeze[(e + 1u) / rw] != 0) { a1.p = p.y; a1.m = m.y; a1.v = v.y; }
            if (row_freeze[(e + 2u) / rw] != 0) { a2.p = p.z; a2.m = m.z; a2.v = v.z; }
            if (row_freeze[(e + 3u) / rw] != 0) { a3.p = p.w; a3.m = m.w; a3.v = v.w; }
        }
        ((float4*)M)[i] = make_float4(a0.m, a1.m, a2.m, a3.m);
        ((float4*)V)[i] = make_float4(a0.v, a1.v, a2.v, a3.v);
        ((float4*)P)[i] = make_float4(a0.p, a1.p, a2.p, a3.p);
    }
    for (long long j = 4 * nvec + (long long)blockIdx.x * 256 + threadIdx.x; j < numel; j += stride) {
        if (rw && row_freeze[(unsigned)j / rw] != 0) continue;
        const AdamPMV a0 = adam_update(P[j], M[j], V[j], G[j], beta1, beta2, eps, inv_bc2_sqrt, step_size);
        M[j] = a0.m; V[j] = a0.v; P[j] = a0.p;
    }
}
// STEP BUMP (capturable path, round 5): `done` is a zeroed device word; every workgroup counts itself in when it has finished, and the one that
// completes the count advances the step counter (or, under a tripped guard, the skipped-steps counter) and re-zeroes the word.  Every workgroup
// reads the step count at its START, the bump happens after ALL of them have finished: no workgroup can see the advanced value — and the
// one-thread adam_bump_step_kernel (3.9 us of launch floor per iteration) is gone.
template <bool CAPTURABLE>
__global__ __launch_bounds__(256) void adam_tensor_kernel(AdamTable t, const double* __restrict__ lr_dev, const int* step_dev,
                                                          float beta1, float beta2, float eps, float inv_bc2_sqrt_host,
                                                          const unsigned* __restrict__ guard_count, unsigned guard_limit,
                                                          const int* __restrict__ live_rows, int* step_rw, unsigned* done, unsigned* skipped_dev,
                                                          const int* __restrict__ row_freeze, const int* __restrict__ grad_rows) {
    __shared__ double s_bc1;
    __shared__ float s_inv_bc2_sqrt;
    const bool skip = CAPTURABLE && guard_count && *guard_count > guard_limit;   // grid-uniform (one scalar load)
    if (!skip) adam_tensor_body<CAPTURABLE>(t, lr_dev, step_dev, beta1, beta2, eps, inv_bc2_sqrt_host, live_rows, &s_bc1, &s_inv_bc2_sqrt, row_freeze, grad_rows);
    if (CAPTURABLE && done) {
        __syncthreads();
        if (threadIdx.x == 0) {      // no fence: only the counter and the step word are shared, and a workgroup reads the step word before it counts itself in
            if (atomicAdd(done, 1u) == gridDim.x * gridDim.y - 1u) {
                *done = 0u;
                if (skip) { if (skipped_dev) *skipped_dev += 1u; } else *step_rw += 1;
            }
        }
    }
}
__global__ void adam_bump_step_kernel(int* step_dev, const unsigned* __restrict__ guard_count, unsigned guard_limit, unsigned* skipped_dev) {
    if (guard_count && *guard_count > guard_limit) {
        if (skipped_dev) *skipped_dev += 1u;
        return;
    }
    *step_dev += 1;
}

// ------------------------------------------------------------------------------------------------ activations
// GaussianModel's getters [REF scene/gaussian_model.py:105-125, 44-56]: opacity = sigmoid(_opacity), scaling = exp(_scaling),
// rotation = normalize(_rotation) (torch.nn.functional.normalize: x / max(||x||, 1e-12)).  One thread per Gaussian, one
// launch for the three of them (the torch chain is ~8 launches forward and ~12 backward at ~5 us each).
__global__ __launch_bounds__(256) void activations_forward_kernel(int P, const float* __restrict__ o_raw, const float* __restrict__ s_raw,
                                                                  const float4* __restrict__ q_raw, float* __restrict__ o,
                                                                  float* __restrict__ s, float4* __restrict__ q, const int* __restrict__ live_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || (live_rows && i >= *live_rows)) return;     // rows behind the live count are not Gaussians: outputs left alone
    o[i] = 1.f / (1.f + expf(-o_raw[i]));
#pragma unroll
    for (int d = 0; d < 3; ++d) s[3 * (size_t)i + d] = expf(s_raw[3 * (size_t)i + d]);
    const float4 r = q_raw[i];
    const float n = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
    q[i] = make_float4(r.x / n, r.y / n, r.z / n, r.w / n);
}
// d/d_raw: sigmoid' = o (1 - o);  exp' = s;  normalize: (g - y (y . g)) / n  with y = x / n  (n above the clamp; at the clamp g / 1e-12)
__global__ __launch_bounds__(256) void activations_backward_kernel(int P, const float* __restrict__ o, const float* __restrict__ s,
                                                                   const float4* __restrict__ q_raw, const float* __restrict__ g_o,
                                                                   const float* __restrict__ g_s, const float4* __restrict__ g_q,
                                                                   float* __restrict__ d_o_raw, float* __restrict__ d_s_raw,
                                                                   float4* __restrict__ d_q_raw, const int* __restrict__ live_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (live_rows && i >= *live_rows) {                        // not a Gaussian: zero gradient (its activations were never computed)
        if (d_o_raw) d_o_raw[i] = 0.f;
        if (d_s_raw) { d_s_raw[3 * (size_t)i] = 0.f; d_s_raw[3 * (size_t)i + 1] = 0.f; d_s_raw[3 * (size_t)i + 2] = 0.f; }
        if (d_q_raw) d_q_raw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    if (d_o_raw) { const float ov = o[i]; d_o_raw[i] = g_o ? g_o[i] * ov * (1.f - ov) : 0.f; }
    if (d_s_raw) {
#pragma unroll
        for (int d = 0; d < 3; ++d) d_s_raw[3 * (size_t)i + d] = g_s ? g_s[3 * (size_t)i + d] * s[3 * (size_t)i + d] : 0.f;
    }
    if (d_q_raw) {
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_q) {
            const float4 r = q_raw[i], g = g_q[i];
            const float nn = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
            if (nn > 1e-12f) {
                const float inv = 1.f / nn;
                const float yx = r.x * inv, yy = r.y * inv, yz = r.z * inv, yw = r.w * inv;
                const float dot = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
                out = make_float4((g.x - yx * dot) * inv, (g.y - yy * dot) * inv, (g.z - yz * dot) * inv, (g.w - yw * dot) * inv);
            } else {
                out = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
            }
        }
        d_q_raw[i] = out;
    }
}


// ---- multi-GPU mapper: the gradient rows of the visible Gaussians, packed for ONE all-reduce -----------------------------------
// Every rank preprocesses all Gaussians, so radii (and with them the packed order: ascending Gaussian index) are identical everywhere; the
// packed block has a static size (row_capacity rows + one flag word), so the all-reduce can sit inside a captured hipGraph.
constexpr int PACK_MAX_ARRAYS = 8;
struct PackTable {
    float* arr[PACK_MAX_ARRAYS];     // (P, width) row-major gradient tensors
    int width[PACK_MAX_ARRAYS];
    int offset[PACK_MAX_ARRAYS];     // column of the array's first float inside a packed row
    int n_arrays, row_floats;
};
__global__ __launch_bounds__(256) void rows_count_kernel(int P, const int* __restrict__ radii, unsigned* __restrict__ block_count) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long b = __ballot(i < P && radii[i] > 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// PACK = true: rows -> packed (and the local overflow flag); false: packed -> rows of the visible Gaussians (and the global flag).
template <bool PACK>
__global__ __launch_bounds__(256) void rows_move_kernel(int P, const int* __restrict__ radii, const unsigned* __restrict__ block_base,
                                                        const int* __restrict__ n_vis, PackTable t, float* __restrict__ packed, int row_capacity,
                                                        const unsigned* __restrict__ guard_count, unsigned guard_limit,
                                                        unsigned* __restrict__ overflow_out) {
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* flag = packed + (size_t)row_capacity * t.row_floats;
    if (i == 0) {
        if (PACK) *flag = (*n_vis > row_capacity || (guard_count && *guard_count > guard_limit)) ? 1.f : 0.f;
        else if (overflow_out) *overflow_out = *flag > 0.f ? 1u : 0u;      // after the all-reduce: some rank overflowed
    }
    if (PACK) {   // rows behind the visible count are never read back, but they ARE summed in place by every all-reduce: keep them zero
        const int nv = *n_vis;
        for (long long z = (long long)nv + i; z < row_capacity; z += (long long)gridDim.x * 256) {
            float* __restrict__ zr = packed + (size_t)z * t.row_floats;
            for (int c = 0; c < t.row_floats; ++c) zr[c] = 0.f;
        }
    }
    const bool k = i < P && radii[i] > 0;
    const unsigned long long b = __ballot(k);
    if (lane == 0) s_w[wave] = (unsigned)__popcll(b);
    __syncthreads();
    if (!k) return;
    unsigned pos = block_base[blockIdx.x] + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    if (pos >= (unsigned)row_capacity) return;                              // overflow: flagged above, the optimiser step is skipped
    float* __restrict__ row = packed + (size_t)pos * t.row_floats;
    for (int a = 0; a < t.n_arrays; ++a) {
        const int w = t.width[a];
        float* __restrict__ g = t.arr[a] + (size_t)i * w;
        float* __restrict__ r = row + t.offset[a];
        for (int c = 0; c < w; ++c) { if (PACK) r[c] = g[c]; else g[c] = r[c]; }
    }
}


// ---- multi-GPU mapper: a rank's own 16x16 tiles (dealt in 2x2 super-tiles: raster_common.hpp tile_xy_is_mine) as one contiguous all-gather chunk
// chunk layout: [slot k][channel: r, g, b, depth][256 pixels of the tile, row-major]; slot k of rank r = tile (k & 3) of its (k >> 2)-th
// super-tile S = (k >> 2) * tile_mod + r; slots past the image edge and pixels outside the image are zero.
template <bool PACK>
__global__ __launch_bounds__(256) void tiles_move_kernel(int W, int H, int gx, int gy, int tile_mod, int tile_rem, int chunk_slots,
                                                         float* __restrict__ color, float* __restrict__ depth, float* __restrict__ buf) {
    // PACK: blockIdx.x = slot k of this rank.  UNPACK: blockIdx.x = tile t, read from the chunk of the rank that owns it.
    const int sgx = (gx + 1) >> 1, sgy = (gy + 1) >> 1;
    int tx, ty, k, src_rank;
    bool valid;
    if (PACK) {
        k = (int)blockIdx.x; src_rank = 0;
        const int S = (k >> 2) * tile_mod + tile_rem;
        tx = 2 * (S % sgx) + (k & 1); ty = 2 * (S / sgx) + ((k >> 1) & 1);
        valid = S < sgx * sgy && tx < gx && ty < gy;
    } else {
        const int t = (int)blockIdx.x;
        tx = t % gx; ty = t / gx;
        const int S = (ty >> 1) * sgx + (tx >> 1);
        src_rank = S % tile_mod;
        k = (S / tile_mod) * 4 + ((ty & 1) * 2 + (tx & 1));
        valid = true;
    }
    float* __restrict__ slab = buf + ((size_t)src_rank * chunk_slots + k) * 1024 + threadIdx.x;
    const int x = tx * 16 + (threadIdx.x & 15), y = ty * 16 + (threadIdx.x >> 4);
    const bool inside = valid && x < W && y < H;
    const size_t HW = (size_t)W * H, pix = (size_t)y * W + x;
    if (PACK) {
        slab[0] = inside ? color[pix] : 0.f;
        slab[256] = inside ? color[HW + pix] : 0.f;
        slab[512] = inside ? color[2 * HW + pix] : 0.f;
        slab[768] = inside ? depth[pix] : 0.f;
    } else if (inside) {
        color[pix] = slab[0]; color[HW + pix] = slab[256]; color[2 * HW + pix] = slab[512]; depth[pix] = slab[768];
    }
}

}  // namespace
}  // namespace gsicp

using namespace gsicp;

static std::atomic<int>& loss_hoist_flag() {
    static std::atomic<int> v([] { const char* e = getenv("GSICP_LOSS_HOIST"); return (e && e[0] == '0') ? 0 : 1; }());
    return v;
}

extern "C" {

int gsicp_mapper_loss_set_hoist(int hoist) { return loss_hoist_flag().exchange(hoist ? 1 : 0); }

size_t gsicp_mapper_loss_scratch_bytes(int width, int height) {
    const size_t HW = (size_t)width * height;
    const size_t tiles = (size_t)((width + LT - 1) / LT) * ((height + LT - 1) / LT);
    return align_up(9 * HW * sizeof(float)) + align_up(4 * tiles * 2 * sizeof(float));
}

static int mapper_loss_impl(const float* image, const float* depth, const float* gt_image, const float* gt_depth, int width, int height,
                            float lambda_dssim, float depth_weight, float d_max, float* loss_out, float* dL_dimage, float* dL_ddepth,
                            char* scratch, int tile_mod, int tile_rem, void* stream_v, const float* const* gt_slots = nullptr,
                            int* bump_step = nullptr, const unsigned* bump_guard = nullptr, unsigned bump_limit = 0u, unsigned* bump_skipped = nullptr) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) { g_last_error = "gsicp_mapper_loss_sharded: bad tile_mod / tile_rem"; return -2; }
    if (width <= 0 || height <= 0 || !image || !depth || (!gt_slots && (!gt_image || !gt_depth)) || !loss_out || !scratch) {
        g_last_error = "gsicp_mapper_loss: bad arguments"; return -2;
    }
    Win win;
    {   // the reference's window [REF utils/loss_utils.py:27-35]: g = exp(-(x-5)^2 / (2*1.5^2)) as float32, g / g.sum(),
        // then the float32 outer product g g^T (torch.mm of an (11,1) by a (1,11) tensor = one rounded multiply per entry)
        float g1[11];
        double sum_d = 0.0;
        for (int i = 0; i < 11; ++i) g1[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        // torch's float32 .sum() of these 11 values equals the correctly rounded exact sum (a plain sequential float32 sum is one
        // ulp lower, and that 6e-8 is visible in SSIM); tests/test_oracle_loss.py checks the construction bit-for-bit
        for (int i = 0; i < 11; ++i) sum_d += (double)g1[i];
        const float sum = (float)sum_d;
        for (int i = 0; i < 11; ++i) win.w[i] = g1[i] / sum;
    }
    const size_t HW = (size_t)width * height;
    // GSICP_LOSS_DEPTH_IN_CH0=1: the two passes with grid z = 3, a tile's depth term computed by its channel-0 workgroup — measured and NOT adopted
    // (round 5: 32.4 + 31.6 us against 31.9 + 30.7: the z = 3 slice's workgroups leave their CU slots at once; they were not costing workgroup rounds)
    static const bool depth_slice = [] { const char* e = getenv("GSICP_LOSS_DEPTH_IN_CH0"); return !(e && e[0] == '1'); }();
    const dim3 grid((width + LT - 1) / LT, (height + LT - 1) / LT, 4);                          // the sharded (fused) kernel keeps its own depth slice
    const dim3 grid2p(grid.x, grid.y, depth_slice ? 4 : 3);                                    // the two-pass form: depth folded into channel 0
    const int n_tiles = grid.x * grid.y;
    float* abc = (float*)scratch;
    float* partial = (float*)(scratch + align_up(9 * HW * sizeof(float)));
    const float n_img = 3.f * (float)HW;
    const LossReduceArgs red{(const float2*)partial, n_tiles, 1.f / n_img, 1.f / (float)HW, lambda_dssim, depth_weight, loss_out, lambda_dssim,
                             bump_step, bump_guard, bump_limit, bump_skipped};
    const bool with_grads = dL_dimage && dL_ddepth;
    if (tile_mod > 1) {
        // Multi-GPU (tile_mod ranks): THIS rank's 32x32 blocks only, in ONE kernel — the fused form needs no derivative maps of blocks other
        // ranks own (the two-pass form would: its second pass reads them with a 5-pixel halo), so its work divides by the rank count:
        // 70 us / N against 33 + 33 us replicated (on one GPU the two passes are faster: csrc/experiments/README.md).  loss_out receives this
        // rank's SHARE {loss, L1, SSIM mean, depth L1}: the sum over the ranks is the loss (the constant lambda is split N ways).
        const LossReduceArgs red_n{(const float2*)partial, n_tiles, 1.f / n_img, 1.f / (float)HW, lambda_dssim, depth_weight, loss_out,
                                   lambda_dssim / (float)tile_world(tile_mod, tile_rem), bump_step, bump_guard, bump_limit, bump_skipped};
        { ProfileScope ps(ST_LOSS_PASS1, stream);
          hipLaunchKernelGGL(loss_fused_kernel, grid, dim3(256), 0, stream, image, depth, gt_image, gt_depth, width, height, win, d_max,
                             -lambda_dssim / n_img, (1.f - lambda_dssim) / n_img, depth_weight / ((float)HW * d_max), dL_dimage, dL_ddepth, partial,
                             tile_mod, tile_rem, gt_slots); }
        { ProfileScope ps(ST_LOSS_PASS2, stream);
          hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, red_n); }
        if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_loss_sharded: kernel launch failed"; return -1; }
        return 0;
    }
    // round 6: pass 2 requests the pixel values it needs AFTER its convolutions together with its staging loads (GSICP_LOSS_HOIST=0 / gsicp_mapper_loss_set_hoist:
    // the kernel of rounds 2-5, which ended every workgroup with an exposed round trip to memory: 30.9 -> 27.8 us).  Same bits either way.
    const bool hoist = loss_hoist_flag().load() != 0;
    { ProfileScope ps(ST_LOSS_PASS1, stream);
      hipLaunchKernelGGL(loss_pass1_kernel, grid2p, dim3(256), 0, stream, image, depth, gt_image, gt_depth, width, height, win, d_max,
                         -lambda_dssim / n_img, abc, partial, gt_slots);
      if (!with_grads) hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, red); }
    if (with_grads) {   // one workgroup of pass 2 finishes the loss value (no launch of its own)
        ProfileScope ps(ST_LOSS_PASS2, stream);
        if (hoist)
            hipLaunchKernelGGL(loss_pass2_kernel<true>, grid2p, dim3(256), 0, stream, image, depth, gt_image, gt_depth, width, height, win, d_max,
                               (1.f - lambda_dssim) / n_img, depth_weight / ((float)HW * d_max), abc, dL_dimage, dL_ddepth, red, gt_slots);
        else
            hipLaunchKernelGGL(loss_pass2_kernel<false>, grid2p, dim3(256), 0, stream, image, depth, gt_image, gt_depth, width, height, win, d_max,
                               (1.f - lambda_dssim) / n_img, depth_weight / ((float)HW * d_max), abc, dL_dimage, dL_ddepth, red, gt_slots);
    }
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_loss: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_loss(const float* image, const float* depth, const float* gt_image, const float* gt_depth, int width, int height,
                      float lambda_dssim, float depth_weight, float d_max, float* loss_out, float* dL_dimage, float* dL_ddepth,
                      char* scratch, void* stream) {
    return mapper_loss_impl(image, depth, gt_image, gt_depth, width, height, lambda_dssim, depth_weight, d_max, loss_out, dL_dimage, dL_ddepth,
                            scratch, 1, 0, stream);
}
int gsicp_mapper_loss_sharded(const float* image, const float* depth, const float* gt_image, const float* gt_depth, int width, int height,
                              float lambda_dssim, float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out,
                              float* dL_dimage, float* dL_ddepth, char* scratch, void* stream) {
    return mapper_loss_impl(image, depth, gt_image, gt_depth, width, height, lambda_dssim, depth_weight, d_max, loss_out, dL_dimage, dL_ddepth,
                            scratch, tile_mod, tile_rem, stream);
}

int gsicp_mapper_loss_indirect(const float* image, const float* depth, const float* const* gt_slots, int width, int height, float lambda_dssim,
                               float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out, float* dL_dimage, float* dL_ddepth,
                               char* scratch, void* stream) {
    if (!gt_slots) { g_last_error = "gsicp_mapper_loss_indirect: gt_slots is NULL"; return -2; }
    return mapper_loss_impl(image, depth, nullptr, nullptr, width, height, lambda_dssim, depth_weight, d_max, loss_out, dL_dimage, dL_ddepth, scratch,
                            tile_mod < 1 ? 1 : tile_mod, tile_rem, stream, gt_slots);
}

int gsicp_mapper_loss_indirect_bump(const float* image, const float* depth, const float* const* gt_slots, int width, int height, float lambda_dssim,
                                    float depth_weight, float d_max, int tile_mod, int tile_rem, float* loss_out, float* dL_dimage, float* dL_ddepth,
                                    char* scratch, int* step_dev, const unsigned int* guard_count, unsigned int guard_limit, unsigned int* skipped_dev,
                                    void* stream) {
    if (!gt_slots || !step_dev) { g_last_error = "gsicp_mapper_loss_indirect_bump: gt_slots / step_dev is NULL"; return -2; }
    return mapper_loss_impl(image, depth, nullptr, nullptr, width, height, lambda_dssim, depth_weight, d_max, loss_out, dL_dimage, dL_ddepth, scratch,
                            tile_mod < 1 ? 1 : tile_mod, tile_rem, stream, gt_slots, step_dev, guard_count, guard_limit, skipped_dev);
}

size_t gsicp_store_compact_scratch_bytes(int n) { return ((size_t)(n > 0 ? (n + 255) / 256 : 1) + 1) * sizeof(unsigned); }

int gsicp_store_compact(int n, const unsigned char* keep, int n_arrays, const void* const* src, void* const* dst, const int* row_bytes,
                        void* scratch, int* n_out_dev, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (n < 0 || n_arrays < 0 || n_arrays > COMPACT_MAX_ARRAYS || !n_out_dev || (n > 0 && (!keep || !scratch))) {
        g_last_error = "gsicp_store_compact: bad arguments (at most 24 arrays)"; return -2;
    }
    CompactTable t;
    t.n_arrays = n_arrays;
    for (int a = 0; a < COMPACT_MAX_ARRAYS; ++a) { t.src[a] = nullptr; t.dst[a] = nullptr; t.row_words[a] = 0; }
    for (int a = 0; a < n_arrays; ++a) {
        if (row_bytes[a] <= 0 || (row_bytes[a] & 3) || !src[a] || !dst[a] || src[a] == dst[a]) {
            g_last_error = "gsicp_store_compact: rows must be non-empty multiples of 4 bytes and src != dst"; return -2;
        }
        t.src[a] = (const unsigned*)src[a]; t.dst[a] = (unsigned*)dst[a]; t.row_words[a] = row_bytes[a] / 4;
    }
    const int nblocks = n > 0 ? (n + 255) / 256 : 1;
    unsigned* block_count = (unsigned*)scratch;
    if (n == 0) {
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, 0, block_count, n_out_dev);
    } else {
        hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, stream, n, keep, block_count);
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, nblocks, block_count, n_out_dev);
        hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblocks), dim3(256), 0, stream, n, keep, (const unsigned*)block_count, t);
    }
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_store_compact: kernel launch failed"; return -1; }
    return 0;
}

size_t gsicp_rows_pack_scratch_bytes(int P) { return ((size_t)(P > 0 ? (P + 255) / 256 : 1) + 2) * sizeof(unsigned); }

static int rows_table(const char* who, int P, const int* radii, int n_arrays, float* const* arrays, const int* row_width, float* packed,
                      int row_capacity, void* scratch, PackTable& t) {
    if (P <= 0 || !radii || n_arrays <= 0 || n_arrays > PACK_MAX_ARRAYS || !arrays || !row_width || !packed || row_capacity <= 0 || !scratch) {
        g_last_error = std::string(who) + ": bad arguments (1..8 arrays, P > 0, row_capacity > 0)"; return -2;
    }
    t.n_arrays = n_arrays; t.row_floats = 0;
    for (int a = 0; a < PACK_MAX_ARRAYS; ++a) { t.arr[a] = nullptr; t.width[a] = 0; t.offset[a] = 0; }
    for (int a = 0; a < n_arrays; ++a) {
        if (!arrays[a] || row_width[a] <= 0) { g_last_error = std::string(who) + ": null array or non-positive row width"; return -2; }
        t.arr[a] = arrays[a]; t.width[a] = row_width[a]; t.offset[a] = t.row_floats; t.row_floats += row_width[a];
    }
    return 0;
}

int gsicp_rows_pack(int P, const int* radii, int n_arrays, const float* const* src, const int* row_width, float* packed, int row_capacity,
                    const unsigned int* guard_count, unsigned int guard_limit, void* scratch, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    PackTable t;
    if (int rc = rows_table("gsicp_rows_pack", P, radii, n_arrays, (float* const*)src, row_width, packed, row_capacity, scratch, t)) return rc;
    const int nblocks = (P + 255) / 256;
    unsigned* block_count = (unsigned*)scratch;
    int* n_vis = (int*)(block_count + nblocks);
    hipLaunchKernelGGL(rows_count_kernel, dim3(nblocks), dim3(256), 0, stream, P, radii, block_count);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, nblocks, block_count, n_vis);
    hipLaunchKernelGGL(rows_move_kernel<true>, dim3(nblocks), dim3(256), 0, stream, P, radii, (const unsigned*)block_count, (const int*)n_vis, t,
                       packed, row_capacity, guard_count, guard_limit, (unsigned*)nullptr);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_rows_pack: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_rows_unpack(int P, const int* radii, int n_arrays, float* const* dst, const int* row_width, const float* packed, int row_capacity,
                      const void* scratch, unsigned int* overflow_out, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    PackTable t;
    if (int rc = rows_table("gsicp_rows_unpack", P, radii, n_arrays, dst, row_width, (float*)packed, row_capacity, (void*)scratch, t)) return rc;
    const int nblocks = (P + 255) / 256;
    const unsigned* block_base = (const unsigned*)scratch;
    hipLaunchKernelGGL(rows_move_kernel<false>, dim3(nblocks), dim3(256), 0, stream, P, radii, block_base, (const int*)(block_base + nblocks), t,
                       (float*)packed, row_capacity, (const unsigned*)nullptr, 0u, overflow_out);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_rows_unpack: kernel launch failed"; return -1; }
    return 0;
}

size_t gsicp_tiles_chunk_floats(int width, int height, int tile_mod) {
    if (width <= 0 || height <= 0 || tile_mod <= 0) return 0;
    return (size_t)tile_chunk_slots((width + 15) / 16, (height + 15) / 16, tile_mod) * 1024;
}

int gsicp_tiles_pack(int width, int height, int tile_mod, int tile_rem, const float* color, const float* depth, float* chunk, void* stream) {
    if (width <= 0 || height <= 0 || tile_mod <= 0 || tile_rem < 0 || tile_rem >= tile_mod || !color || !depth || !chunk) {
        g_last_error = "gsicp_tiles_pack: bad arguments"; return -2;
    }
    const int gx = (width + 15) / 16, gy = (height + 15) / 16, chunk_slots = tile_chunk_slots(gx, gy, tile_mod);
    hipLaunchKernelGGL(tiles_move_kernel<true>, dim3(chunk_slots), dim3(256), 0, (hipStream_t)stream, width, height, gx, gy, tile_mod, tile_rem,
                       chunk_slots, (float*)color, (float*)depth, chunk);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_tiles_pack: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_tiles_unpack(int width, int height, int tile_mod, const float* gathered, float* color, float* depth, void* stream) {
    if (width <= 0 || height <= 0 || tile_mod <= 0 || !gathered || !color || !depth) { g_last_error = "gsicp_tiles_unpack: bad arguments"; return -2; }
    const int gx = (width + 15) / 16, gy = (height + 15) / 16, chunk_slots = tile_chunk_slots(gx, gy, tile_mod);
    hipLaunchKernelGGL(tiles_move_kernel<false>, dim3(gx * gy), dim3(256), 0, (hipStream_t)stream, width, height, gx, gy, tile_mod, 0, chunk_slots, color,
                       depth, (float*)gathered);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_tiles_unpack: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_set_view(int width, int height, const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image,
                          const float* gt_depth, float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, float* dst_gt_image,
                          float* dst_gt_depth, void* stream) {
    const size_t HW = (size_t)width * height;
    if (width <= 0 || height <= 0 || (HW & 3) || !viewmatrix || !projmatrix || !campos || !gt_image || !gt_depth || !dst_viewmatrix ||
        !dst_projmatrix || !dst_campos || !dst_gt_image || !dst_gt_depth) {
        g_last_error = "gsicp_mapper_set_view: bad arguments (width*height must be a multiple of 4)"; return -2;
    }
    if ((((uintptr_t)gt_image | (uintptr_t)gt_depth | (uintptr_t)dst_gt_image | (uintptr_t)dst_gt_depth) & 15) != 0) {
        g_last_error = "gsicp_mapper_set_view: image buffers must be 16-byte aligned"; return -2;
    }
    hipLaunchKernelGGL(set_view_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, viewmatrix, projmatrix, campos, (const float4*)gt_image,
                       (const float4*)gt_depth, 3 * HW / 4, HW / 4, dst_viewmatrix, dst_projmatrix, dst_campos, (float4*)dst_gt_image,
                       (float4*)dst_gt_depth);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_set_view: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_select_view(const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image, const float* gt_depth,
                             float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, const float** dst_gt_slots, void* stream) {
    if (!viewmatrix || !projmatrix || !campos || !gt_image || !gt_depth || !dst_viewmatrix || !dst_projmatrix || !dst_campos || !dst_gt_slots) {
        g_last_error = "gsicp_mapper_select_view: bad arguments"; return -2;
    }
    hipLaunchKernelGGL(select_view_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, viewmatrix, projmatrix, campos, gt_image, gt_depth, dst_viewmatrix,
                       dst_projmatrix, dst_campos, dst_gt_slots);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_select_view: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_select_view_zero(const float* viewmatrix, const float* projmatrix, const float* campos, const float* gt_image, const float* gt_depth,
                                  float* dst_viewmatrix, float* dst_projmatrix, float* dst_campos, const float** dst_gt_slots, void* zero_region,
                                  size_t zero_words, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (!viewmatrix || !projmatrix || !campos || !gt_image || !gt_depth || !dst_viewmatrix || !dst_projmatrix || !dst_campos || !dst_gt_slots || (zero_words && !zero_region)) {
        g_last_error = "gsicp_mapper_select_view_zero: NULL argument"; return -2;
    }
    if (zero_words > (size_t)1 << 22) { g_last_error = "gsicp_mapper_select_view_zero: zero region above 16 MB (not a rasteriser counter region)"; return -2; }
    hipLaunchKernelGGL(select_view_zero_kernel, dim3(1), dim3(256), 0, stream, viewmatrix, projmatrix, campos, gt_image, gt_depth, dst_viewmatrix, dst_projmatrix,
                       dst_campos, dst_gt_slots, (uint32_t*)zero_region, zero_words);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_select_view_zero: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_activations_forward(int P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                                     float* scaling, float* rotation, const int* live_rows_dev, void* stream) {
    if (P < 0 || (P > 0 && (!opacity_raw || !scaling_raw || !rotation_raw || !opacity || !scaling || !rotation))) {
        g_last_error = "gsicp_mapper_activations_forward: bad arguments"; return -2;
    }
    if (P == 0) return 0;
    hipLaunchKernelGGL(activations_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, opacity_raw, scaling_raw,
                       (const float4*)rotation_raw, opacity, scaling, (float4*)rotation, live_rows_dev);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_activations_forward: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_mapper_activations_backward(int P, const float* opacity, const float* scaling, const float* rotation_raw, const float* dL_dopacity,
                                      const float* dL_dscaling, const float* dL_drotation, float* dL_dopacity_raw, float* dL_dscaling_raw,
                                      float* dL_drotation_raw, const int* live_rows_dev, void* stream) {
    if (P < 0 || (P > 0 && (!opacity || !scaling || !rotation_raw))) { g_last_error = "gsicp_mapper_activations_backward: bad arguments"; return -2; }
    if (P == 0) return 0;
    hipLaunchKernelGGL(activations_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, opacity, scaling,
                       (const float4*)rotation_raw, dL_dopacity, dL_dscaling, (const float4*)dL_drotation, dL_dopacity_raw, dL_dscaling_raw,
                       (float4*)dL_drotation_raw, live_rows_dev);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_mapper_activations_backward: kernel launch failed"; return -1; }
    return 0;
}

// GSICP_ADAM_INKERNEL_BUMP=1: the step bump by the last workgroup of the Adam launch instead of the one-thread bump kernel (an experiment: the first
// version, with an agent-scope fence per workgroup — on gfx950 a write-back of the XCD's L2 — took the Adam kernel from 20 to 188 us; default OFF)
static std::atomic<int> g_adam_bump_kernel([] { const char* e = getenv("GSICP_ADAM_INKERNEL_BUMP"); return (e && e[0] == '1') ? 0 : 1; }());
static unsigned* adam_done_word(const int* step_dev, hipStream_t stream) {
    static std::mutex mu;
    static std::map<const int*, unsigned*> words;     // keyed by the step counter: one optimiser = one counter = one word
    std::lock_guard<std::mutex> lk(mu);
    auto it = words.find(step_dev);
    if (it != words.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    unsigned* w = nullptr;
    if (hipMalloc((void**)&w, 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(w, 0, 256) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); return nullptr; }
    if (words.size() > 4096) words.clear();           // step counters of long-gone optimisers (the words themselves are 256 B each: left allocated)
    words[step_dev] = w;
    return w;
}

static long long adam_table(AdamTable& t, int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const long long* numel, const float* lr, double bc1) {
    long long total = 0;
    t.n = 0;
    t.step_offset = 1;
    for (int k = 0; k < n_groups; ++k) {
        if (numel[k] <= 0) continue;
        t.p[t.n] = params[k]; t.g[t.n] = grads[k]; t.m[t.n] = exp_avg[k]; t.v[t.n] = exp_avg_sq[k];
        total += numel[k];
        t.end[t.n] = total;
        t.step_size[t.n] = lr ? (float)((double)lr[k] / bc1) : 0.f;
        t.src[t.n] = k;
        t.row_width[t.n] = 0;
        t.frozen[t.n] = 0;
        ++t.n;
    }
    for (int k = t.n; k < ADAM_MAX_GROUPS; ++k) {
        t.p[k] = nullptr; t.g[k] = nullptr; t.m[k] = nullptr; t.v[k] = nullptr; t.end[k] = total; t.step_size[k] = 0.f; t.src[k] = 0;
        t.row_width[k] = 0; t.frozen[k] = 0;
    }
    return total;
}

static dim3 adam_grid(const AdamTable& t) {   // x: 16-byte units of the largest tensor (grid-stride beyond 1024 blocks), y: tensor
    long long largest = 0;
    for (int k = 0; k < t.n; ++k) { const long long ne = t.end[k] - (k ? t.end[k - 1] : 0); largest = ne > largest ? ne : largest; }
    long long blocks = (largest / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    return dim3((unsigned)blocks, (unsigned)t.n);
}

int gsicp_adam_step(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const long long* numel, const float* lr, float beta1, float beta2, float eps, int step, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (n_groups < 0 || n_groups > ADAM_MAX_GROUPS || step < 1) { g_last_error = "gsicp_adam_step: 1..8 tensors, step >= 1"; return -2; }
    AdamTable t;
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const long long total = adam_table(t, n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr, bc1);
    if (total == 0) return 0;
    hipLaunchKernelGGL(adam_tensor_kernel<false>, adam_grid(t), dim3(256), 0, stream, t, (const double*)nullptr, (const int*)nullptr, beta1, beta2, eps,
                       (float)(1.0 / std::sqrt(bc2)), (const unsigned*)nullptr, 0u, (const int*)nullptr, (int*)nullptr, (unsigned*)nullptr,
                       (unsigned*)nullptr, (const int*)nullptr, (const int*)nullptr);
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_adam_step: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_adam_step_guarded(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                            float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                            unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, void* stream_v) {
    return gsicp_adam_step_masked(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, beta1, beta2, eps, step_dev, bump_step, guard_count,
                                  guard_limit, skipped_dev, live_rows_dev, row_width, nullptr, nullptr, stream_v);
}

int gsicp_adam_step_masked(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                           float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                           unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, const int* row_freeze_dev,
                           const int* group_frozen, void* stream_v) {
    return gsicp_adam_step_sparse(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, beta1, beta2, eps, step_dev, bump_step, guard_count,
                                  guard_limit, skipped_dev, live_rows_dev, row_width, row_freeze_dev, group_frozen, nullptr, stream_v);
}

int gsicp_adam_step_sparse(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                           float eps, int* step_dev, int bump_step, const unsigned int* guard_count, unsigned int guard_limit,
                           unsigned int* skipped_dev, const int* live_rows_dev, const int* row_width, const int* row_freeze_dev,
                           const int* group_frozen, const int* grad_rows_dev, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (grad_rows_dev && !row_width) { g_last_error = "gsicp_adam_step_sparse: grad_rows needs row_width"; return -2; }
    if (bump_step < 0 || bump_step > 2) { g_last_error = "gsicp_adam_step_sparse: bump_step is 0 (none), 1 (bump after the update) or 2 (already bumped)"; return -2; }
    if (row_freeze_dev && (!row_width || !group_frozen)) { g_last_error = "gsicp_adam_step_masked: row_freeze needs row_width and group_frozen"; return -2; }
    if (n_groups < 0 || n_groups > ADAM_MAX_GROUPS || !lr_dev || !step_dev) {
        g_last_error = "gsicp_adam_step_capturable: 1..8 tensors, device lr array and device step counter"; return -2;
    }
    AdamTable t;
    const long long total = adam_table(t, n_groups, params, grads, exp_avg, exp_avg_sq, numel, nullptr, 1.0);
    if ((live_rows_dev || row_freeze_dev || grad_rows_dev) && row_width)
        for (int k = 0; k < t.n; ++k) t.row_width[k] = row_width[t.src[k]];
    // bump_step 2: the step counter was advanced BEFORE this launch, inside the same stream order (gsicp_mapper_loss_indirect_bump): use it as it is
    t.step_offset = bump_step == 2 ? 0 : 1;
    if (bump_step == 2) bump_step = 0;
    if (row_freeze_dev)
        for (int k = 0; k < t.n; ++k) t.frozen[k] = group_frozen[t.src[k]] ? 1 : 0;
    // the word the workgroups of the bumping launch count themselves into (see adam_tensor_kernel): one per (device, step counter), zeroed once.
    // It cannot be allocated while a stream is being captured; a capture always follows eager warm-up steps, which allocate it.  Without it
    // (or with nothing to update) the one-thread bump kernel runs, as in rounds 2-4.
    unsigned* done = (bump_step && total > 0 && !g_adam_bump_kernel.load()) ? adam_done_word(step_dev, stream) : nullptr;
    {
        ProfileScope ps(ST_ADAM, stream);
        if (total > 0)
            hipLaunchKernelGGL(adam_tensor_kernel<true>, adam_grid(t), dim3(256), 0, stream, t, lr_dev, (const int*)step_dev, beta1, beta2, eps, 0.f,
                               guard_count, guard_limit, (live_rows_dev && row_width) ? live_rows_dev : (const int*)nullptr, step_dev, done, skipped_dev,
                               row_freeze_dev, grad_rows_dev);
        if (bump_step && !done)
            hipLaunchKernelGGL(adam_bump_step_kernel, dim3(1), dim3(1), 0, stream, step_dev, guard_count, guard_limit, skipped_dev);
    }
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_adam_step_capturable: kernel launch failed"; return -1; }
    return 0;
}

int gsicp_adam_step_capturable(int n_groups, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float beta1, float beta2,
                               float eps, int* step_dev, void* stream_v) {
    return gsicp_adam_step_guarded(n_groups, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, beta1, beta2, eps, step_dev, 1, nullptr, 0u,
                                   nullptr, nullptr, nullptr, stream_v);
}

}  // extern "C"
