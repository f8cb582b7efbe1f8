// Per-Gaussian kernels of the gfx950 rasteriser: forward preprocess (R1), its backward (R8+R9) and mark_visible (R10).
//
// Replaces the per-Gaussian stage of diff_gaussian_rasterization reached from
// GaussianRasterizer.forward [REF gaussian_renderer/__init__.py:294-302] / loss.backward() [REF mp_Mapper.py:242].
// Quaternions are (x,y,z,w) [REF utils/general_utils.py:89-99]; matrices are row-vector form
// [REF scene/shared_objs.py:163-166].
//
// This translation unit is compiled with -ffp-contract=off: view-space depth (the sort key), the pixel centre,
// the radius and the tile rectangle feed INTEGER outputs that the parity tests compare bit-exactly, so every
// float expression here is evaluated left-to-right without FMA fusion.  One thread per Gaussian, 256-thread
// blocks (4 waves); the work is 56 B read + ~66 B written per Gaussian and embarrassingly parallel, so the
// kernel is HBM/launch bound.  Camera matrices are wave-uniform and live in SGPRs.
#include "raster_common.hpp"
#include <atomic>
#include <cstdlib>

namespace gsicp {

namespace {

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Cam {
    float v[16];
    float p[16];
};

__device__ inline void load_cam(const float* view, const float* proj, Cam& c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { c.v[i] = view[i]; c.p[i] = proj[i]; }
}

__device__ inline void quat_to_R(const float* q, float* Rm) {
    const float x = q[0], y = q[1], z = q[2], r = q[3];
    Rm[0] = 1.f - 2.f * (y * y + z * z);
    Rm[1] = 2.f * (x * y - r * z);
    Rm[2] = 2.f * (x * z + r * y);
    Rm[3] = 2.f * (x * y + r * z);
    Rm[4] = 1.f - 2.f * (x * x + z * z);
    Rm[5] = 2.f * (y * z - r * x);
    Rm[6] = 2.f * (x * z - r * y);
    Rm[7] = 2.f * (y * z + r * x);
    Rm[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ inline void cov3_from_scale_rot(const float* s, float mod, const float* q, float* c6) {
    float Rm[9];
    quat_to_R(q, Rm);
    const float s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
    float L[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        L[3 * i + 0] = Rm[3 * i + 0] * s0;
        L[3 * i + 1] = Rm[3 * i + 1] * s1;
        L[3 * i + 2] = Rm[3 * i + 2] * s2;
    }
#define GS_DOT(i, j) (L[3 * (i)] * L[3 * (j)] + L[3 * (i) + 1] * L[3 * (j) + 1] + L[3 * (i) + 2] * L[3 * (j) + 2])
    c6[0] = GS_DOT(0, 0); c6[1] = GS_DOT(0, 1); c6[2] = GS_DOT(0, 2);
    c6[3] = GS_DOT(1, 1); c6[4] = GS_DOT(1, 2); c6[5] = GS_DOT(2, 2);
#undef GS_DOT
}

// 2x3 matrix M = J * Rw2c with J the perspective Jacobian at the (frustum-clamped) view-space point.
__device__ inline void ewa_M(const float* t_in, float fx, float fy, float tanx, float tany, const float* view, float* Mm,
                             float* tcl, bool* clx, bool* cly) {
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t_in[0] / t_in[2], tytz = t_in[1] / t_in[2];
    const float cx = fminf(limx, fmaxf(-limx, txtz));
    const float cy = fminf(limy, fmaxf(-limy, tytz));
    if (clx) *clx = (txtz < -limx || txtz > limx);
    if (cly) *cly = (tytz < -limy || tytz > limy);
    const float tx = cx * t_in[2], ty = cy * t_in[2], tz = t_in[2];
    if (tcl) { tcl[0] = tx; tcl[1] = ty; tcl[2] = tz; }
    const float j00 = fx / tz, j02 = -(fx * tx) / (tz * tz);
    const float j11 = fy / tz, j12 = -(fy * ty) / (tz * tz);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Mm[k] = j00 * view[4 * k + 0] + j02 * view[4 * k + 2];
        Mm[3 + k] = j11 * view[4 * k + 1] + j12 * view[4 * k + 2];
    }
}

__device__ inline void cov2_from_M(const float* Mm, const float* c6, float* abc) {
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float v0[3], v1[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v0[i] = S[3 * i] * Mm[0] + S[3 * i + 1] * Mm[1] + S[3 * i + 2] * Mm[2];
        v1[i] = S[3 * i] * Mm[3] + S[3 * i + 1] * Mm[4] + S[3 * i + 2] * Mm[5];
    }
    abc[0] = (Mm[0] * v0[0] + Mm[1] * v0[1] + Mm[2] * v0[2]) + 0.3f;
    abc[1] = Mm[0] * v1[0] + Mm[1] * v1[1] + Mm[2] * v1[2];
    abc[2] = (Mm[3] * v1[0] + Mm[4] * v1[1] + Mm[5] * v1[2]) + 0.3f;
}

__device__ inline void sh_to_rgb(int deg, const float* mean, const float* campos, const float* sh, float* rgb, unsigned* clampmask) {
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = SH_C0 * sh[c];
    if (deg > 0) {
        const float dx = mean[0] - campos[0], dy = mean[1] - campos[1], dz = mean[2] - campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            res[c] = res[c] - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                res[c] = res[c] + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] +
                         SH_C2[2] * (2.f * zz - xx - yy) * sh[18 + c] + SH_C2[3] * xz * sh[21 + c] +
                         SH_C2[4] * (xx - yy) * sh[24 + c];
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    res[c] = res[c] + SH_C3[0] * y * (3.f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                             SH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + c] +
                             SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
                             SH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                             SH_C3[6] * x * (xx - 3.f * yy) * sh[45 + c];
            }
        }
    }
    unsigned m = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        res[c] = res[c] + 0.5f;
        if (res[c] < 0.f) m |= (1u << c);
        rgb[c] = fmaxf(res[c], 0.f);
    }
    *clampmask = m;
}

// GaussianModel's activation getters [REF scene/gaussian_model.py:44-56, 105-125], applied in place of a separate launch when the caller hands
// over the raw parameters (raw_params): the same formulas as activations_forward_kernel / torch (sigmoid, exp, x / max(||x||, 1e-12)).
__device__ inline float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ inline float act_normalize4(const float* r, float* y) {
    const float n = fmaxf(sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]), 1e-12f);
    y[0] = r[0] / n; y[1] = r[1] / n; y[2] = r[2] / n; y[3] = r[3] / n;
    return n;
}

__global__ __launch_bounds__(256) void preprocess_kernel(PreprocessArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool in_range = idx < a.P;
    const bool live = in_range && (a.live_rows == nullptr || idx < *a.live_rows);   // rows behind the live count are culled Gaussians
    uint32_t mine = 0;    // emission slots this Gaussian needs (tiles of this rank it touches)
    if (in_range) {
    Cam cam;
    load_cam(a.view, a.proj, cam);
    const int gx = (a.W + TILE - 1) / TILE, gy = (a.H + TILE - 1) / TILE;
    const float fx = (float)a.W / (2.f * a.tanfovx), fy = (float)a.H / (2.f * a.tanfovy);

    // defaults for a culled Gaussian
    a.radii[idx] = 0;
    a.clamped[idx] = 0;
    if (a.is_used_zero) a.is_used_zero[idx] = 0;
    SplatRec rec = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const float p[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float pv[3];
    pv[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
    pv[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
    pv[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
    bool ok = live && pv[2] > 0.2f;
    if (ok) {
        const float ph0 = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
        const float ph1 = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
        const float ph3 = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
        const float pw = 1.f / (ph3 + 0.0000001f);
        const float ndcx = ph0 * pw, ndcy = ph1 * pw;
        float c6[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = a.cov3D_precomp[6 * idx + k];
        } else {
            float s[3] = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
            float q[4] = {a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2], a.rotations[4 * idx + 3]};
            if (a.raw_params) {
                s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
                const float raw[4] = {q[0], q[1], q[2], q[3]};
                (void)act_normalize4(raw, q);
            }
            cov3_from_scale_rot(s, a.scale_modifier, q, c6);
        }
        float Mm[6], abc[3];
        ewa_M(pv, fx, fy, a.tanfovx, a.tanfovy, cam.v, Mm, nullptr, nullptr, nullptr);
        cov2_from_M(Mm, c6, abc);
        const float det = abc[0] * abc[2] - abc[1] * abc[1];
        ok = det != 0.f;
        if (ok) {
            const float det_inv = 1.f / det;
            rec.ca = abc[2] * det_inv; rec.cb = -abc[1] * det_inv; rec.cc = abc[0] * det_inv;
            const float mid = 0.5f * (abc[0] + abc[2]);
            const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float l1 = mid + disc, l2 = mid - disc;
            const float radf = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
            rec.px = ((ndcx + 1.f) * (float)a.W - 1.f) * 0.5f;
            rec.py = ((ndcy + 1.f) * (float)a.H - 1.f) * 0.5f;
            const int rad = (int)radf;
            int x0, y0, x1, y1;
            tile_rect(rec.px, rec.py, rad, gx, gy, x0, y0, x1, y1);
            const int ntiles = (x1 - x0) * (y1 - y0);
            ok = ntiles != 0;
            if (ok) {
                unsigned cm = 0;
                float rgb[3];
                if (a.colors_precomp) {
                    rgb[0] = a.colors_precomp[3 * idx]; rgb[1] = a.colors_precomp[3 * idx + 1]; rgb[2] = a.colors_precomp[3 * idx + 2];
                } else {
                    sh_to_rgb(a.D, p, a.campos, a.shs + (size_t)3 * a.M * idx, rgb, &cm);
                }
                rec.r = rgb[0]; rec.g = rgb[1]; rec.b = rgb[2];
                rec.depth = pv[2];
                rec.opacity = a.raw_params ? act_sigmoid(a.opacities[idx]) : a.opacities[idx];
                // Footprint of alpha = min(0.99, o * exp(power)) >= 1/255:  q(d) = -2 power <= 2 ln(255 o).  Its bounding
                // box has half-extents sqrt(2 tau * Sigma2_xx), sqrt(2 tau * Sigma2_yy).  tau is padded by 0.01 (1 % in
                // alpha) and the box by half a pixel, far more than any fp32 / __expf rounding, so culling with it never
                // drops a pixel the exact per-pixel test would accept.
                const float tau = logf(255.f * rec.opacity) + 0.01f;
                if (tau > 0.f) {
                    rec.hx = sqrtf(2.f * tau * abc[0]) + 0.5f;
                    rec.hy = sqrtf(2.f * tau * abc[2]) + 0.5f;
                } else {
                    rec.hx = -1e30f; rec.hy = -1e30f;
                }
                mine = (uint32_t)ntiles;
                if (a.tile_mod > 1) {  // multi-GPU tile sharding: only this rank's tiles get emission slots
                    mine = 0;
                    for (int y = y0; y < y1; ++y)
                        for (int x = x0; x < x1; ++x) mine += tile_xy_is_mine(x, y, gx, a.tile_mod, a.tile_rem) ? 1u : 0u;
                }
                a.radii[idx] = rad;
                a.clamped[idx] = (unsigned char)cm;
            }
        }
    }
    if (ok) a.rec[idx] = rec;     // a culled Gaussian's record is never read (radii == 0, no emission slots): 48 B x ~3/4 of the map not written
    }  // in_range
    // Emission-slot allocation: each Gaussian gets a private contiguous run of `mine` slots.  The runs need no global
    // order (only contiguity), so one block-aggregated atomic per workgroup replaces a device-wide prefix scan.
    // Round 5: the same atomic also allocates the workgroup's share of the backward's WORK LIST — the ids of the Gaussians that survived the culls
    // (radii > 0), compacted by ballot rank.  The two counters are adjacent 32-bit words ([0] = slots, [1] = visible Gaussians) of one 8-byte-aligned
    // 64-bit word: ONE 64-bit atomicAdd returns both bases (a second atomic doubled the kernel's dependent chain: +7 us, measured).  The list's
    // order across workgroups is whatever the atomics gave; nothing depends on it (every Gaussian's gradients are computed independently).
    __shared__ uint32_t s_wave[4], s_vwave[4];
    __shared__ unsigned long long s_base64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    const bool vis = a.vis_list != nullptr && in_range && a.radii[idx] > 0;
    const unsigned long long vm = __ballot(vis);
    const uint32_t vrank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_vwave[wave] = (uint32_t)__popcll(vm);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const uint32_t vtot = s_vwave[0] + s_vwave[1] + s_vwave[2] + s_vwave[3];
        unsigned long long base = 0ull;
        if (tot | vtot) {
            if (a.vis_list) base = atomicAdd((unsigned long long*)a.total_counter, (unsigned long long)tot | ((unsigned long long)vtot << 32));
            else base = (unsigned long long)atomicAdd(a.total_counter, tot);
        }
        s_base64 = base;
    }
    __syncthreads();
    const uint32_t s_base = (uint32_t)s_base64, s_vbase = (uint32_t)(s_base64 >> 32);
    uint32_t wave_off = 0, voff = 0;
    for (int w = 0; w < wave; ++w) { wave_off += s_wave[w]; voff += s_vwave[w]; }
    if (in_range) {
        a.tiles_touched[idx] = mine;
        a.slot_base[idx] = s_base + wave_off + incl - mine;
    }
    if (vis) a.vis_list[s_vbase + voff + vrank] = (uint32_t)idx;
}

// ------------------------------------------------------------------------------------------------ backward
// Round 5: the backward functions below are compiled WITHOUT FMA contraction, like the rest of this file (rounds 1-4 allowed it here).  With
// contraction the compiler chooses which a * b + c become FMAs per kernel it inlines this code into: the legacy kernel and the compacted kernel
// of round 5 then rounded differently in the last bit, and one ill-conditioned Gaussian of the S-map crossed the parity bound.  Without it every
// operation is one IEEE operation in source order — the two kernels are bit-identical by construction (tests/test_raster_gpu.py), and the sequence
// is the one the fp32 oracle evaluates.  The kernel is latency-bound (VALU issue floor 4.6 us of 12): the extra multiplies cost nothing.

__device__ inline void sh_backward(int deg, int M, const float* mean, const float* campos, const float* sh, unsigned cm,
                                   const float* dcol, float* dL_dsh, float* dm) {
    float dRGB[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dRGB[c] = (cm >> c) & 1u ? 0.f : dcol[c];
    for (int k = 3; k < 3 * M; ++k) dL_dsh[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dL_dsh[c] = SH_C0 * dRGB[c];
    if (deg == 0) return;
    const float dir[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    float dx[3], dy[3], dz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dL_dsh[3 + c] = -SH_C1 * y * dRGB[c];
        dL_dsh[6 + c] = SH_C1 * z * dRGB[c];
        dL_dsh[9 + c] = -SH_C1 * x * dRGB[c];
        dx[c] = -SH_C1 * sh[9 + c];
        dy[c] = -SH_C1 * sh[3 + c];
        dz[c] = SH_C1 * sh[6 + c];
    }
    if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dL_dsh[12 + c] = SH_C2[0] * xy * dRGB[c];
            dL_dsh[15 + c] = SH_C2[1] * yz * dRGB[c];
            dL_dsh[18 + c] = SH_C2[2] * (2.f * zz - xx - yy) * dRGB[c];
            dL_dsh[21 + c] = SH_C2[3] * xz * dRGB[c];
            dL_dsh[24 + c] = SH_C2[4] * (xx - yy) * dRGB[c];
            dx[c] += SH_C2[0] * y * sh[12 + c] + SH_C2[2] * 2.f * -x * sh[18 + c] + SH_C2[3] * z * sh[21 + c] + SH_C2[4] * 2.f * x * sh[24 + c];
            dy[c] += SH_C2[0] * x * sh[12 + c] + SH_C2[1] * z * sh[15 + c] + SH_C2[2] * 2.f * -y * sh[18 + c] + SH_C2[4] * 2.f * -y * sh[24 + c];
            dz[c] += SH_C2[1] * y * sh[15 + c] + SH_C2[2] * 2.f * 2.f * z * sh[18 + c] + SH_C2[3] * x * sh[21 + c];
        }
        if (deg > 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dL_dsh[27 + c] = SH_C3[0] * y * (3.f * xx - yy) * dRGB[c];
                dL_dsh[30 + c] = SH_C3[1] * xy * z * dRGB[c];
                dL_dsh[33 + c] = SH_C3[2] * y * (4.f * zz - xx - yy) * dRGB[c];
                dL_dsh[36 + c] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * dRGB[c];
                dL_dsh[39 + c] = SH_C3[4] * x * (4.f * zz - xx - yy) * dRGB[c];
                dL_dsh[42 + c] = SH_C3[5] * z * (xx - yy) * dRGB[c];
                dL_dsh[45 + c] = SH_C3[6] * x * (xx - 3.f * yy) * dRGB[c];
                dx[c] += SH_C3[0] * sh[27 + c] * 3.f * 2.f * xy + SH_C3[1] * sh[30 + c] * yz + SH_C3[2] * sh[33 + c] * -2.f * xy +
                         SH_C3[3] * sh[36 + c] * -3.f * 2.f * xz + SH_C3[4] * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) +
                         SH_C3[5] * sh[42 + c] * 2.f * xz + SH_C3[6] * sh[45 + c] * 3.f * (xx - yy);
                dy[c] += SH_C3[0] * sh[27 + c] * 3.f * (xx - yy) + SH_C3[1] * sh[30 + c] * xz +
                         SH_C3[2] * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[36 + c] * -3.f * 2.f * yz +
                         SH_C3[4] * sh[39 + c] * -2.f * xy + SH_C3[5] * sh[42 + c] * -2.f * yz + SH_C3[6] * sh[45 + c] * -3.f * 2.f * xy;
                dz[c] += SH_C3[1] * sh[30 + c] * xy + SH_C3[2] * sh[33 + c] * 4.f * 2.f * yz +
                         SH_C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[39 + c] * 4.f * 2.f * xz +
                         SH_C3[5] * sh[42 + c] * (xx - yy);
            }
        }
    }
    float dd[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) { dd[0] += dx[c] * dRGB[c]; dd[1] += dy[c] * dRGB[c]; dd[2] += dz[c] * dRGB[c]; }
    const float sum2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    const float inv = 1.f / sqrtf(sum2 * sum2 * sum2);
    dm[0] += ((sum2 - dir[0] * dir[0]) * dd[0] - dir[1] * dir[0] * dd[1] - dir[2] * dir[0] * dd[2]) * inv;
    dm[1] += (-dir[0] * dir[1] * dd[0] + (sum2 - dir[1] * dir[1]) * dd[1] - dir[2] * dir[1] * dd[2]) * inv;
    dm[2] += (-dir[0] * dir[2] * dd[0] - dir[1] * dir[2] * dd[1] + (sum2 - dir[2] * dir[2]) * dd[2]) * inv;
}

// Per-Gaussian algebra of the backward (R8 / R9) for Gaussian i.  gs = moment sums of h = G dL/dalpha over every pixel this Gaussian blends
// (zeros for an invisible one); r_early / p_early / sc_early / q_early = its splat record and parameters, loaded by the caller next to the sums.
__device__ __forceinline__ void prebwd_finish(const PreprocessBwdArgs& a, const int i, const bool visible, float* gs, const SplatRec& r_early,
                                              const float* p_early, const float* sc_early, const float* q_early) {
    float dm[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dsc[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    const bool use_sh = (a.colors_precomp == nullptr) && (a.dL_dsh != nullptr);
    float sc_act[3] = {1.f, 1.f, 1.f}, q_act[4] = {0.f, 0.f, 0.f, 1.f}, q_norm_v = 1.f;   // activated scales / quaternion and the raw norm (raw_params)
    // gs = moment sums of h = G dL/dalpha over every pixel this Gaussian blends: [h dx, h dy, h dx dx, h dx dy, h dy dy, h, ...]
    // (blend_backward_tile_kernel); the per-Gaussian constants turn them into the screen-space gradients
    if (visible) {
        const SplatRec r = r_early;
        const float hx = gs[0], hy = gs[1], hxx = gs[2], hxy = gs[3], hyy = gs[4];
        gs[0] = r.opacity * (-hx * r.ca - hy * r.cb) * (0.5f * (float)a.W);
        gs[1] = r.opacity * (-hy * r.cc - hx * r.cb) * (0.5f * (float)a.H);
        gs[2] = -0.5f * hxx * r.opacity;
        gs[3] = -hxy * r.opacity;
        gs[4] = -0.5f * hyy * r.opacity;
    }
    a.dL_dmean2D[3 * i] = gs[0]; a.dL_dmean2D[3 * i + 1] = gs[1]; a.dL_dmean2D[3 * i + 2] = 0.f;
    // dL_dconic, dL_ddepths, dL_dcolors (when SHs are used) and dL_dcov3D (when scales / rotations are used) are intermediate results the caller
    // only needs for inspection: a NULL pointer skips their 56 B per Gaussian of writes (round 3)
    if (a.dL_dconic) { a.dL_dconic[4 * i] = gs[2]; a.dL_dconic[4 * i + 1] = gs[3]; a.dL_dconic[4 * i + 2] = gs[4]; a.dL_dconic[4 * i + 3] = 0.f; }
    a.dL_dopacity[i] = (a.raw_params && visible) ? gs[5] * r_early.opacity * (1.f - r_early.opacity) : gs[5];   // sigmoid' = o (1 - o)
    if (a.dL_dcolors) { a.dL_dcolors[3 * i] = gs[6]; a.dL_dcolors[3 * i + 1] = gs[7]; a.dL_dcolors[3 * i + 2] = gs[8]; }
    if (a.dL_ddepths) a.dL_ddepths[i] = gs[9];
    if (visible) {
        Cam cam;
        load_cam(a.view, a.proj, cam);
        const float fx = (float)a.W / (2.f * a.tanfovx), fy = (float)a.H / (2.f * a.tanfovy);
        const float p[3] = {p_early[0], p_early[1], p_early[2]};
        float c6[6];
        float q[4] = {0.f, 0.f, 0.f, 1.f}, sc[3] = {1.f, 1.f, 1.f};
        float q_raw[4] = {0.f, 0.f, 0.f, 1.f}, q_norm = 1.f;
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = a.cov3D_precomp[6 * i + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) sc[k] = sc_early[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = q_early[k];
            if (a.raw_params) {
#pragma unroll
                for (int k = 0; k < 3; ++k) sc[k] = expf(sc[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) q_raw[k] = q[k];
                q_norm = act_normalize4(q_raw, q);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) sc_act[k] = sc[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) q_act[k] = q[k];
            q_norm_v = sqrtf(q_raw[0] * q_raw[0] + q_raw[1] * q_raw[1] + q_raw[2] * q_raw[2] + q_raw[3] * q_raw[3]);
            (void)q_norm;
            cov3_from_scale_rot(sc, a.scale_modifier, q, c6);
        }
        float pv[3], Mm[6], tcl[3], abc[3];
        bool clx, cly;
        pv[0] = cam.v[0] * p[0] + cam.v[4] * p[1] + cam.v[8] * p[2] + cam.v[12];
        pv[1] = cam.v[1] * p[0] + cam.v[5] * p[1] + cam.v[9] * p[2] + cam.v[13];
        pv[2] = cam.v[2] * p[0] + cam.v[6] * p[1] + cam.v[10] * p[2] + cam.v[14];
        ewa_M(pv, fx, fy, a.tanfovx, a.tanfovy, cam.v, Mm, tcl, &clx, &cly);
        cov2_from_M(Mm, c6, abc);
        const float A = abc[0], B = abc[1], C = abc[2];
        const float det = A * C - B * B;
        const float gA = gs[2], gB = gs[3], gC = gs[4];
        const float d2inv = 1.f / (det * det + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (d2inv != 0.f) {
            dL_da = d2inv * (-C * C * gA + B * C * gB + (det - A * C) * gC);
            dL_dc = d2inv * (-A * A * gC + A * B * gB + (det - A * C) * gA);
            dL_db = d2inv * (2.f * B * C * gA - (det + 2.f * B * B) * gB + 2.f * A * B * gC);
        }
        const float h = 0.5f * dL_db;
        float GM[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) { GM[k] = dL_da * Mm[k] + h * Mm[3 + k]; GM[3 + k] = h * Mm[k] + dL_dc * Mm[3 + k]; }
#define GS_FULL(r, qq) (Mm[r] * GM[qq] + Mm[3 + (r)] * GM[3 + (qq)])
        dcov[0] = GS_FULL(0, 0); dcov[3] = GS_FULL(1, 1); dcov[5] = GS_FULL(2, 2);
        dcov[1] = 2.f * GS_FULL(0, 1); dcov[2] = 2.f * GS_FULL(0, 2); dcov[4] = 2.f * GS_FULL(1, 2);
#undef GS_FULL
        const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        float dM[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) dM[3 * r + k] = 2.f * (GM[3 * r] * S[k] + GM[3 * r + 1] * S[3 + k] + GM[3 * r + 2] * S[6 + k]);
#define GS_DJ(r, col) (dM[3 * (r)] * cam.v[col] + dM[3 * (r) + 1] * cam.v[4 + (col)] + dM[3 * (r) + 2] * cam.v[8 + (col)])
        const float dJ00 = GS_DJ(0, 0), dJ02 = GS_DJ(0, 2), dJ11 = GS_DJ(1, 1), dJ12 = GS_DJ(1, 2);
#undef GS_DJ
        const float tz = 1.f / tcl[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float xm = clx ? 0.f : 1.f, ym = cly ? 0.f : 1.f;
        const float dtx = xm * (-fx * tz2 * dJ02);
        const float dty = ym * (-fy * tz2 * dJ12);
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * tcl[0]) * tz3 * dJ02 + (2.f * fy * tcl[1]) * tz3 * dJ12;
#pragma unroll
        for (int k = 0; k < 3; ++k) dm[k] += cam.v[4 * k] * dtx + cam.v[4 * k + 1] * dty + cam.v[4 * k + 2] * dtz;
        const float gd = gs[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) dm[k] += cam.v[4 * k + 2] * gd;
        {
            const float ph0 = cam.p[0] * p[0] + cam.p[4] * p[1] + cam.p[8] * p[2] + cam.p[12];
            const float ph1 = cam.p[1] * p[0] + cam.p[5] * p[1] + cam.p[9] * p[2] + cam.p[13];
            const float ph3 = cam.p[3] * p[0] + cam.p[7] * p[1] + cam.p[11] * p[2] + cam.p[15];
            const float mw = 1.f / (ph3 + 0.0000001f);
            const float mul1 = ph0 * mw * mw, mul2 = ph1 * mw * mw;
            const float g0 = gs[0], g1 = gs[1];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dm[k] += (cam.p[4 * k] * mw - cam.p[4 * k + 3] * mul1) * g0 + (cam.p[4 * k + 1] * mw - cam.p[4 * k + 3] * mul2) * g1;
        }
        if (use_sh) {
            const float dcol[3] = {gs[6], gs[7], gs[8]};
            sh_backward(a.D, a.M, p, a.campos, a.shs + (size_t)3 * a.M * i, a.clamped[i], dcol, a.dL_dsh + (size_t)3 * a.M * i, dm);
        }
        if (!a.cov3D_precomp) {
            float Rm[9];
            quat_to_R(q, Rm);
            const float sv[3] = {a.scale_modifier * sc[0], a.scale_modifier * sc[1], a.scale_modifier * sc[2]};
            const float G3[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                                 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
            float dLm[9], dR[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    dLm[3 * r + k] = 2.f * (G3[3 * r] * Rm[k] + G3[3 * r + 1] * Rm[3 + k] + G3[3 * r + 2] * Rm[6 + k]) * sv[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) { acc += dLm[3 * r + k] * Rm[3 * r + k]; dR[3 * r + k] = dLm[3 * r + k] * sv[k]; }
                dsc[k] = a.scale_modifier * acc;
            }
            const float x = q[0], y = q[1], z = q[2], r = q[3];
            const float d00 = dR[0], d01 = dR[1], d02 = dR[2], d10 = dR[3], d11 = dR[4], d12 = dR[5], d20 = dR[6], d21 = dR[7], d22 = dR[8];
            drot[0] = 2.f * (y * (d01 + d10) + z * (d02 + d20) + r * (d21 - d12)) - 4.f * x * (d11 + d22);
            drot[1] = 2.f * (x * (d01 + d10) + z * (d12 + d21) + r * (d02 - d20)) - 4.f * y * (d00 + d22);
            drot[2] = 2.f * (x * (d02 + d20) + y * (d12 + d21) + r * (d10 - d01)) - 4.f * z * (d00 + d11);
            drot[3] = 2.f * (x * (d21 - d12) + y * (d02 - d20) + z * (d10 - d01));
        }
    } else if (use_sh) {
        for (int k = 0; k < 3 * a.M; ++k) a.dL_dsh[(size_t)3 * a.M * i + k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) a.dL_dmeans3D[3 * i + k] = dm[k];
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = dcov[k];
    }
    if (a.raw_params && visible && !a.cov3D_precomp) {
        // chain rule of the activations (as activations_backward_kernel): exp' = s;  normalize: (g - y (y . g)) / n above the 1e-12 clamp
#pragma unroll
        for (int k = 0; k < 3; ++k) dsc[k] *= sc_act[k];
        const float dot = q_act[0] * drot[0] + q_act[1] * drot[1] + q_act[2] * drot[2] + q_act[3] * drot[3];
        if (q_norm_v > 1e-12f) {
            const float inv = 1.f / q_norm_v;
#pragma unroll
            for (int k = 0; k < 4; ++k) drot[k] = (drot[k] - q_act[k] * dot) * inv;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) drot[k] *= 1e12f;
        }
    }
    if (a.dL_dscales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dL_dscales[3 * i + k] = dsc[k];
    }
    if (a.dL_drots) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.dL_drots[4 * i + k] = drot[k];
    }
}

__device__ __forceinline__ void prebwd_load(const PreprocessBwdArgs& a, const int i, SplatRec& r, float* p, float* sc, float* q) {
    r = a.rec[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = a.means3D[3 * i + k];
    if (!a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sc[k] = a.scales[3 * i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = a.rotations[4 * i + k];
    }
}

// LEGACY per-Gaussian pass (rounds 3-4; GSICP_PREBWD_LEGACY=1, kept as the A/B partner and the reference of tests/test_raster_gpu.py): one thread
// per Gaussian over all P, each visible thread walks its own run of 48-byte records, left to right.
__global__ __launch_bounds__(256) void preprocess_backward_legacy_kernel(PreprocessBwdArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const bool visible = a.radii[i] > 0 && (a.live_rows == nullptr || i < *a.live_rows);
    float gs[NGRAD];
#pragma unroll
    for (int c = 0; c < NGRAD; ++c) gs[c] = 0.f;
    SplatRec r_early;
    r_early.px = r_early.py = r_early.depth = r_early.hx = r_early.ca = r_early.cb = r_early.cc = r_early.opacity = 0.f;
    r_early.r = r_early.g = r_early.b = r_early.hy = 0.f;
    float p_early[3] = {0.f, 0.f, 0.f}, sc_early[3] = {1.f, 1.f, 1.f}, q_early[4] = {0.f, 0.f, 0.f, 1.f};
    if (visible) prebwd_load(a, i, r_early, p_early, sc_early, q_early);
    {
        const uint32_t n_slots = (!visible || *a.total_counter > a.capacity) ? 0u : a.tiles_touched[i];
        const float4* es = (const float4*)a.entry_sum + 3 * (size_t)a.slot_base[i];
        for (uint32_t u0 = 0; u0 < n_slots; u0 += 4) {
            float4 q[4][3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t u = u0 + k < n_slots ? u0 + k : u0;
                q[k][0] = es[3 * u]; q[k][1] = es[3 * u + 1]; q[k][2] = es[3 * u + 2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (u0 + k < n_slots) {
                    gs[0] += q[k][0].x; gs[1] += q[k][0].y; gs[2] += q[k][0].z; gs[3] += q[k][0].w;
                    gs[4] += q[k][1].x; gs[5] += q[k][1].y; gs[6] += q[k][1].z; gs[7] += q[k][1].w;
                    gs[8] += q[k][2].x; gs[9] += q[k][2].y;
                }
            }
        }
    }
    prebwd_finish(a, i, visible, gs, r_early, p_early, sc_early, q_early);
}

// ---- Round 5: the per-Gaussian pass in two launches.
// (1) entry_run_sum_kernel — the SUMMATION of a Gaussian's run of records as a flat, coalesced pass over the emission slots: one wave per window
//     of 64 consecutive slots, one lane per 48-byte record, run boundaries from entry_gauss (slot -> Gaussian id).  The wave stages its 64 records in
//     a wave-private LDS slab; the lane that holds a run's LAST record then adds the run's records from the slab first to last — the same left
//     fold, in slot order, the legacy kernel performed: the totals are BIT-IDENTICAL to rounds 3-4 (and independent of where the slot allocator
//     placed the run).  A run that starts in a window and leaves it is folded by THAT wave, 64 records per trip through the same slab, the
//     accumulator carried across trips; lanes of a run that started in an earlier window idle.  The total replaces the record of the run's last
//     slot.  What the legacy walk paid per Gaussian — a chain of dependent global loads as long as its longest run per wave (342 records on a
//     trained map: 86 rounds) — is now one coalesced load per record plus LDS reads.  (A first version summed with a segmented Hillis-Steele
//     scan across the lanes: 130 ds_bpermute per window bound it to the LDS crossbar, 33 us at D = 1.3 M; and its tree order moved one
//     ill-conditioned Gaussian of the S-map across the parity bound.)
// (2) preprocess_backward_kernel — the per-Gaussian algebra over the forward's COMPACTED list of visible Gaussians (thread t takes list entry t:
//     every lane of the leading waves works; on the S-map 82 % of the Gaussians are culled and the legacy kernel ran its 120-register body with
//     a fifth of the lanes), reading ONE record per Gaussian; thread t also writes the zeros of Gaussian t when that one is invisible.
__global__ __launch_bounds__(256) void entry_run_sum_kernel(const uint32_t* __restrict__ total_counter, const uint32_t capacity,
                                                            const uint32_t* __restrict__ entry_gauss, const uint32_t* __restrict__ tiles_touched,
                                                            float* __restrict__ entry_sum) {
    __shared__ __attribute__((aligned(16))) float4 s_slab[4][64 * 3];
    const uint32_t total = *total_counter;
    const uint32_t R = total > capacity ? 0u : total;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const uint32_t n_windows = (R + 63u) / 64u;
    const uint32_t n_waves = gridDim.x * 4u;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    float4* es = (float4*)entry_sum;
    float4* slab = s_slab[wave];
    for (uint32_t w = blockIdx.x * 4u + (uint32_t)wave; w < n_windows; w += n_waves) {
        const uint32_t u = w * 64u + (uint32_t)lane;
        const bool valid = u < R;
        const uint32_t g = valid ? entry_gauss[u] : NONE;
        const uint32_t g_prev = (valid && u > 0u) ? entry_gauss[u - 1u] : NONE;
        const uint32_t g_next = (u + 1u < R) ? entry_gauss[u + 1u] : NONE;
        const bool head = valid && g != g_prev, tail = valid && g != g_next;
        int hl = head ? lane : -1;               // lane of the head of the run this lane belongs to (-1: the run started in an earlier window)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(hl, off, 64);
            if (lane >= off) hl = o > hl ? o : hl;
        }
        if (valid && hl >= 0) {
            slab[3 * lane] = es[3 * (size_t)u]; slab[3 * lane + 1] = es[3 * (size_t)u + 1]; slab[3 * lane + 2] = es[3 * (size_t)u + 2];
        }
        __builtin_amdgcn_wave_barrier();         // a wave's LDS operations execute in program order; this only stops the compiler from reordering them
        if (tail && hl >= 0) {                   // a complete run inside the window: left fold of slab[hl .. lane], four records' LDS reads in flight
            float v[NGRAD];
#pragma unroll
            for (int c = 0; c < NGRAD; ++c) v[c] = 0.f;
            for (int j = hl; j <= lane; j += 4) {
                float4 q[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u <= lane ? j + u : lane;
                    q[u][0] = slab[3 * jj]; q[u][1] = slab[3 * jj + 1]; q[u][2] = slab[3 * jj + 2];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j + u <= lane) {
                        v[0] += q[u][0].x; v[1] += q[u][0].y; v[2] += q[u][0].z; v[3] += q[u][0].w; v[4] += q[u][1].x; v[5] += q[u][1].y; v[6] += q[u][1].z;
                        v[7] += q[u][1].w; v[8] += q[u][2].x; v[9] += q[u][2].y;
                    }
                }
            }
            es[3 * (size_t)u] = make_float4(v[0], v[1], v[2], v[3]);
            es[3 * (size_t)u + 1] = make_float4(v[4], v[5], v[6], v[7]);
            es[3 * (size_t)u + 2] = make_float4(v[8], v[9], 0.f, 0.f);
        }
        // the run that starts in this window and leaves it: 64 records per trip through the slab; lane c < 12 folds component c (a left fold is
        // a chain of dependent additions: its pace is one addition per record, so the twelve chains run side by side on twelve lanes, sixteen
        // LDS words in flight each — a single lane folding whole records was the kernel's tail: 64 records x 150 cycles per trip)
        const int hl63 = __shfl(hl, 63, 64);
        const int open63 = __shfl((int)(valid && !tail), 63, 64);
        if (open63 != 0 && hl63 >= 0) {
            const uint32_t s0 = w * 64u + (uint32_t)hl63;
            const uint32_t gsp = (uint32_t)__shfl((int)g, 63, 64);
            const uint32_t n = tiles_touched[gsp];
            const float* slab_f = (const float*)slab;
            float acc = 0.f;
            for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
                const uint32_t pos = c0 + (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                if (pos < n) {
                    const size_t uu = (size_t)s0 + pos;
                    slab[3 * lane] = es[3 * uu]; slab[3 * lane + 1] = es[3 * uu + 1]; slab[3 * lane + 2] = es[3 * uu + 2];
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < SLOT_F) {
                    const int cnt = (n - c0) >= 64u ? 64 : (int)(n - c0);
                    for (int j = 0; j < cnt; j += 16) {
                        float x[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) x[u] = slab_f[SLOT_F * (j + u < cnt ? j + u : cnt - 1) + lane];
#pragma unroll
                        for (int u = 0; u < 16; ++u) if (j + u < cnt) acc += x[u];
                    }
                }
            }
            if (lane < SLOT_F) entry_sum[((size_t)s0 + n - 1u) * SLOT_F + lane] = lane < NGRAD ? acc : 0.f;
        }
        __builtin_amdgcn_wave_barrier();         // the next window's stores into the slab stay behind this window's reads
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void preprocess_backward_kernel(PreprocessBwdArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.P) return;
    // SPARSE gradients (a.sparse_grads, round 6: the captured mapper iteration, whose only consumer is FusedAdam with the forward's radii as its row
    // mask): the rows of culled Gaussians and the rows behind the live count of a capacity-backed map are not written at all — on the S-map that is
    // 26 MB of zero stores per launch (82 % of the Gaussians are culled) which the optimiser then read back; at a capacity of 2 M rows and 250 k live
    // ones 120 MB.  Every other caller (autograd owns those tensors: torch.optim.Adam, grad hooks, dense all-reduces) gets every row written.
    SplatRec r0;
    r0.px = r0.py = r0.depth = r0.hx = r0.ca = r0.cb = r0.cc = r0.opacity = 0.f;
    r0.r = r0.g = r0.b = r0.hy = 0.f;
    if (!a.sparse_grads) {   // Gaussian t, if it is not visible: its zeros (the algebra of an invisible Gaussian IS the zero fill)
        const bool visible_t = a.radii[t] > 0 && (a.live_rows == nullptr || t < *a.live_rows);
        if (!visible_t) {
            float gs[NGRAD];
#pragma unroll
            for (int c = 0; c < NGRAD; ++c) gs[c] = 0.f;
            const float p0[3] = {0.f, 0.f, 0.f}, s1[3] = {1.f, 1.f, 1.f}, q1[4] = {0.f, 0.f, 0.f, 1.f};
            prebwd_finish(a, t, false, gs, r0, p0, s1, q1);
        }
    }
    // The forward bumps the slot count (low word) and the visible count (high word) with ONE 64-bit atomic: a duplicate total of 2^32 or more — degenerate
    // near-camera splats over a multi-million-row map; the lists overflowed long before and nothing was rendered — would carry into the visible count
    // (ADVICE r5).  The count is therefore clamped to P and every list entry checked against P before it indexes anything.
    const uint32_t n_vis_raw = *a.vis_counter;
    const uint32_t n_vis = n_vis_raw < (uint32_t)a.P ? n_vis_raw : (uint32_t)a.P;
    if ((uint32_t)t >= n_vis) return;
    const uint32_t iu = a.vis_list[t];
    if (iu >= (uint32_t)a.P) return;
    const int i = (int)iu;
    if (a.live_rows != nullptr && i >= *a.live_rows) return;      // cannot happen (rows behind the live count are culled in the forward); cheap
    SplatRec r_early = r0;
    float p_early[3] = {0.f, 0.f, 0.f}, sc_early[3] = {1.f, 1.f, 1.f}, q_early[4] = {0.f, 0.f, 0.f, 1.f};
    prebwd_load(a, i, r_early, p_early, sc_early, q_early);
    float gs[NGRAD];
#pragma unroll
    for (int c = 0; c < NGRAD; ++c) gs[c] = 0.f;
    const uint32_t n_slots = (*a.total_counter > a.capacity) ? 0u : a.tiles_touched[i];
    if (n_slots != 0u) {
        const float4* es = (const float4*)a.entry_sum + 3 * ((size_t)a.slot_base[i] + n_slots - 1u);     // the run's total (entry_run_sum_kernel)
        const float4 q0 = es[0], q1 = es[1], q2 = es[2];
        gs[0] = q0.x; gs[1] = q0.y; gs[2] = q0.z; gs[3] = q0.w; gs[4] = q1.x; gs[5] = q1.y; gs[6] = q1.z; gs[7] = q1.w; gs[8] = q2.x; gs[9] = q2.y;
    }
    prebwd_finish(a, i, true, gs, r_early, p_early, sc_early, q_early);
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* means3D, const float* view, unsigned char* present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float z = view[2] * means3D[3 * i] + view[6] * means3D[3 * i + 1] + view[10] * means3D[3 * i + 2] + view[14];
    present[i] = z > 0.2f ? 1 : 0;
}

}  // namespace

void launch_preprocess(const PreprocessArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    hipLaunchKernelGGL(preprocess_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}
static std::atomic<int>& prebwd_legacy_flag() {
    static std::atomic<int> v([] { const char* e = getenv("GSICP_PREBWD_LEGACY"); return (e && e[0] == '1') ? 1 : 0; }());
    return v;
}
static bool prebwd_legacy() { return prebwd_legacy_flag().load() != 0; }
int set_prebwd_legacy(int legacy) { return prebwd_legacy_flag().exchange(legacy ? 1 : 0); }
void launch_entry_run_sum(const PreprocessBwdArgs& a, hipStream_t s) {
    if (a.P <= 0 || a.capacity == 0u || prebwd_legacy()) return;
    // grid-stride over the windows of 64 slots that exist on the device (R <= capacity): enough workgroups to fill the chip, no more
    unsigned blocks = (unsigned)(((size_t)a.capacity + 255) / 256);
    if (blocks > 4096u) blocks = 4096u;
    hipLaunchKernelGGL(entry_run_sum_kernel, dim3(blocks), dim3(256), 0, s, a.total_counter, a.capacity, a.entry_gauss, a.tiles_touched, a.entry_sum_rw);
}
void launch_preprocess_backward(const PreprocessBwdArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    if (prebwd_legacy()) hipLaunchKernelGGL(preprocess_backward_legacy_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(preprocess_backward_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
}
void launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

}  // namespace gsicp
