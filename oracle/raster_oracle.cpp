// ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (gs_icp_slam_amd/).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// CPU restatement of the tile-binned, depth-sorted, alpha-blended 3D-Gaussian rasteriser
// (forward + backward, colour + depth) behind `diff_gaussian_rasterization`.
//
// Pinned pieces (golden vectors from the reference's own Python, tests/test_oracle_pinned.py): quaternion convention + 3-D covariance
// assembly (utils/general_utils.py:60-114), SH evaluation degrees 0-3 (utils/sh_utils.py:57-112), camera matrices.
// PARITY UNPINNED: the reference ships submodules/diff-gaussian-rasterization as an EMPTY directory
// (/root/reference/.gitmodules:1-3, no pinned commit) and has no tests or golden vectors, so there is
// no reference file:line for the arithmetic.  This restatement follows
//   (i)  the behavioural constraints visible at the reference's call sites:
//        gaussian_renderer/__init__.py:244-302 (12 settings fields, 8 call kwargs, return order
//        depth,colour,radii,is_used), utils/general_utils.py:89-99 (quaternions are x,y,z,w),
//        scene/shared_objs.py:163-166 (row-vector, pre-transposed matrices), mp_Mapper.py:231-240
//        (depth in metres, compared un-normalised against sensor depth), and
//   (ii) the published algorithm of graphdeco-inria/diff-gaussian-rasterization (the fork's upstream;
//        the viewer pins 3509be80 at SIBR_viewers/src/projects/gaussianviewer/renderer/CMakeLists.txt:13-16):
//        EWA projection with +0.3 px dilation, radius = ceil(3*sqrt(lambda_max)), 16x16 tiles,
//        64-bit (tile<<32 | float-bits(depth)) stable sort, front-to-back blending with
//        alpha = min(0.99, o*exp(power)), skip alpha < 1/255, stop when T would fall below 1e-4.
// Fork-specific choices that cannot be verified here (SURVEY.md §8a "three unknowns"):
//   depth image = sum_i z_i * alpha_i * T_i  (no normalisation, no background term);
//   is_used[i]  = 1 iff Gaussian i passed the alpha and transmittance tests on >= 1 pixel.
//
// The float instantiation evaluates every expression that feeds an INTEGER output (view-space depth
// bits, pixel centre, radius, tile rectangle) in a fixed left-to-right order without FMA contraction
// (build with -ffp-contract=off), so that the HIP preprocess kernel — built the same way — yields
// bit-identical sort keys and index lists.  The double instantiation exists for finite-difference
// gradient checks.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int TILE = 16;

static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154,  -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

template <class R> inline R rexp(R x);
template <> inline float rexp<float>(float x) { return expf(x); }
template <> inline double rexp<double>(double x) { return exp(x); }

template <class R> struct Splat {
    bool visible = false;
    R px = 0, py = 0;          // pixel-space centre
    R depth = 0;               // view-space z
    R ca = 0, cb = 0, cc = 0;  // conic (inverse 2D covariance): power = -.5(ca dx^2 + cc dy^2) - cb dx dy
    R opacity = 0;
    R rgb[3] = {0, 0, 0};
    bool clamped[3] = {false, false, false};
    int radius = 0;
    int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
    R cov3[6] = {0, 0, 0, 0, 0, 0};
};

template <class R> struct Problem {
    int P, D, M, W, H;
    const R *bg, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    R scale_modifier;
    const R *view, *proj, *campos;
    R tanfovx, tanfovy;
    int prefiltered;
};

// Quaternion (x,y,z,w) -> rotation matrix, row-major.  No normalisation (the caller's activation
// normalises; reference scene/gaussian_model.py:117-118).
template <class R> inline void quat_to_R(const R* q, R* Rm) {
    const R x = q[0], y = q[1], z = q[2], r = q[3];
    Rm[0] = R(1) - R(2) * (y * y + z * z);
    Rm[1] = R(2) * (x * y - r * z);
    Rm[2] = R(2) * (x * z + r * y);
    Rm[3] = R(2) * (x * y + r * z);
    Rm[4] = R(1) - R(2) * (x * x + z * z);
    Rm[5] = R(2) * (y * z - r * x);
    Rm[6] = R(2) * (x * z - r * y);
    Rm[7] = R(2) * (y * z + r * x);
    Rm[8] = R(1) - R(2) * (x * x + y * y);
}

// Sigma3 = Rot * diag((mod*s)^2) * Rot^T, stored as (xx, xy, xz, yy, yz, zz).
template <class R> inline void cov3_from_scale_rot(const R* s, R mod, const R* q, R* c6) {
    R Rm[9];
    quat_to_R(q, Rm);
    const R s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
    // L = Rot * S  (column k scaled by s_k)
    R L[9];
    for (int i = 0; i < 3; ++i) {
        L[3 * i + 0] = Rm[3 * i + 0] * s0;
        L[3 * i + 1] = Rm[3 * i + 1] * s1;
        L[3 * i + 2] = Rm[3 * i + 2] * s2;
    }
    auto dot = [&](int i, int j) { return L[3 * i] * L[3 * j] + L[3 * i + 1] * L[3 * j + 1] + L[3 * i + 2] * L[3 * j + 2]; };
    c6[0] = dot(0, 0); c6[1] = dot(0, 1); c6[2] = dot(0, 2);
    c6[3] = dot(1, 1); c6[4] = dot(1, 2); c6[5] = dot(2, 2);
}

template <class R> inline void xform43(const R* p, const R* m, R* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
template <class R> inline void xform44(const R* p, const R* m, R* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// 2x3 matrix Mm = J * Rw2c (J = perspective Jacobian at the clamped view-space point t).
template <class R>
inline void ewa_M(const R* t_in, R fx, R fy, R tanx, R tany, const R* view, R* Mm, R* tcl, bool* clx, bool* cly) {
    const R limx = R(1.3) * tanx, limy = R(1.3) * tany;
    const R txtz = t_in[0] / t_in[2], tytz = t_in[1] / t_in[2];
    const R cx = std::min(limx, std::max(-limx, txtz));
    const R cy = std::min(limy, std::max(-limy, tytz));
    if (clx) *clx = (txtz < -limx || txtz > limx);
    if (cly) *cly = (tytz < -limy || tytz > limy);
    const R tx = cx * t_in[2], ty = cy * t_in[2], tz = t_in[2];
    if (tcl) { tcl[0] = tx; tcl[1] = ty; tcl[2] = tz; }
    const R j00 = fx / tz, j02 = -(fx * tx) / (tz * tz);
    const R j11 = fy / tz, j12 = -(fy * ty) / (tz * tz);
    // Rw2c row i = (view[i], view[4+i], view[8+i])
    for (int k = 0; k < 3; ++k) {
        Mm[k] = j00 * view[4 * k + 0] + j02 * view[4 * k + 2];
        Mm[3 + k] = j11 * view[4 * k + 1] + j12 * view[4 * k + 2];
    }
}

template <class R> inline void cov2_from_M(const R* Mm, const R* c6, R* abc) {
    // V = Sigma3 * M^T rows
    const R S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    R v0[3], v1[3];
    for (int i = 0; i < 3; ++i) {
        v0[i] = S[3 * i] * Mm[0] + S[3 * i + 1] * Mm[1] + S[3 * i + 2] * Mm[2];
        v1[i] = S[3 * i] * Mm[3] + S[3 * i + 1] * Mm[4] + S[3 * i + 2] * Mm[5];
    }
    abc[0] = (Mm[0] * v0[0] + Mm[1] * v0[1] + Mm[2] * v0[2]) + R(0.3);
    abc[1] = Mm[0] * v1[0] + Mm[1] * v1[1] + Mm[2] * v1[2];
    abc[2] = (Mm[3] * v1[0] + Mm[4] * v1[1] + Mm[5] * v1[2]) + R(0.3);
}

template <class R>
inline void sh_to_rgb(int deg, int M, const R* mean, const R* campos, const R* sh /* M x 3 */, R* rgb, bool* clamped) {
    R res[3];
    for (int c = 0; c < 3; ++c) res[c] = R(SH_C0) * sh[c];
    if (deg > 0) {
        R dx = mean[0] - campos[0], dy = mean[1] - campos[1], dz = mean[2] - campos[2];
        const R len = std::sqrt(dx * dx + dy * dy + dz * dz);
        const R x = dx / len, y = dy / len, z = dz / len;
        for (int c = 0; c < 3; ++c)
            res[c] = res[c] - R(SH_C1) * y * sh[3 + c] + R(SH_C1) * z * sh[6 + c] - R(SH_C1) * x * sh[9 + c];
        if (deg > 1) {
            const R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            for (int c = 0; c < 3; ++c)
                res[c] = res[c] + R(SH_C2[0]) * xy * sh[12 + c] + R(SH_C2[1]) * yz * sh[15 + c] +
                         R(SH_C2[2]) * (R(2) * zz - xx - yy) * sh[18 + c] + R(SH_C2[3]) * xz * sh[21 + c] +
                         R(SH_C2[4]) * (xx - yy) * sh[24 + c];
            if (deg > 2) {
                for (int c = 0; c < 3; ++c)
                    res[c] = res[c] + R(SH_C3[0]) * y * (R(3) * xx - yy) * sh[27 + c] + R(SH_C3[1]) * xy * z * sh[30 + c] +
                             R(SH_C3[2]) * y * (R(4) * zz - xx - yy) * sh[33 + c] +
                             R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy) * sh[36 + c] +
                             R(SH_C3[4]) * x * (R(4) * zz - xx - yy) * sh[39 + c] + R(SH_C3[5]) * z * (xx - yy) * sh[42 + c] +
                             R(SH_C3[6]) * x * (xx - R(3) * yy) * sh[45 + c];
            }
        }
    }
    (void)M;
    for (int c = 0; c < 3; ++c) {
        res[c] = res[c] + R(0.5);
        clamped[c] = res[c] < R(0);
        rgb[c] = std::max(res[c], R(0));
    }
}

template <class R> void preprocess(const Problem<R>& pb, std::vector<Splat<R>>& sp) {
    const int gx = (pb.W + TILE - 1) / TILE, gy = (pb.H + TILE - 1) / TILE;
    const R fx = R(pb.W) / (R(2) * pb.tanfovx), fy = R(pb.H) / (R(2) * pb.tanfovy);
    sp.assign(pb.P, Splat<R>());
#pragma omp parallel for schedule(static)
    for (int i = 0; i < pb.P; ++i) {
        Splat<R>& s = sp[i];
        const R* p = pb.means3D + 3 * i;
        R pv[3];
        xform43(p, pb.view, pv);
        if (pv[2] <= R(0.2)) continue;  // frustum cull (near plane only, as upstream)
        R ph[4];
        xform44(p, pb.proj, ph);
        const R pw = R(1) / (ph[3] + R(0.0000001));
        const R ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        if (pb.cov3D_precomp) for (int k = 0; k < 6; ++k) s.cov3[k] = pb.cov3D_precomp[6 * i + k];
        else cov3_from_scale_rot(pb.scales + 3 * i, pb.scale_modifier, pb.rotations + 4 * i, s.cov3);
        R Mm[6], abc[3];
        ewa_M<R>(pv, fx, fy, pb.tanfovx, pb.tanfovy, pb.view, Mm, nullptr, nullptr, nullptr);
        cov2_from_M(Mm, s.cov3, abc);
        const R det = abc[0] * abc[2] - abc[1] * abc[1];
        if (det == R(0)) continue;
        const R det_inv = R(1) / det;
        s.ca = abc[2] * det_inv; s.cb = -abc[1] * det_inv; s.cc = abc[0] * det_inv;
        const R mid = R(0.5) * (abc[0] + abc[2]);
        const R disc = std::sqrt(std::max(R(0.1), mid * mid - det));
        const R l1 = mid + disc, l2 = mid - disc;
        const R radf = std::ceil(R(3) * std::sqrt(std::max(l1, l2)));
        s.px = ((ndcx + R(1)) * R(pb.W) - R(1)) * R(0.5);
        s.py = ((ndcy + R(1)) * R(pb.H) - R(1)) * R(0.5);
        const int rad = (int)radf;
        s.rminx = std::min(gx, std::max(0, (int)((s.px - R(rad)) / R(TILE))));
        s.rminy = std::min(gy, std::max(0, (int)((s.py - R(rad)) / R(TILE))));
        s.rmaxx = std::min(gx, std::max(0, (int)((s.px + R(rad) + R(TILE - 1)) / R(TILE))));
        s.rmaxy = std::min(gy, std::max(0, (int)((s.py + R(rad) + R(TILE - 1)) / R(TILE))));
        if ((s.rmaxx - s.rminx) * (s.rmaxy - s.rminy) == 0) continue;
        if (pb.colors_precomp) { for (int c = 0; c < 3; ++c) s.rgb[c] = pb.colors_precomp[3 * i + c]; }
        else sh_to_rgb<R>(pb.D, pb.M, p, pb.campos, pb.shs + (size_t)3 * pb.M * i, s.rgb, s.clamped);
        s.depth = pv[2];
        s.radius = rad;
        s.opacity = pb.opacities[i];
        s.visible = true;
    }
}

inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// (tile << 32 | depth bits) stable sort of duplicates emitted in Gaussian-index order.
template <class R>
void bin_and_sort(const Problem<R>& pb, const std::vector<Splat<R>>& sp, std::vector<uint64_t>& keys,
                  std::vector<uint32_t>& vals, std::vector<uint32_t>& ranges) {
    const int gx = (pb.W + TILE - 1) / TILE, gy = (pb.H + TILE - 1) / TILE;
    std::vector<std::pair<uint64_t, uint32_t>> kv;
    for (int i = 0; i < pb.P; ++i) {
        const Splat<R>& s = sp[i];
        if (!s.visible) continue;
        for (int y = s.rminy; y < s.rmaxy; ++y)
            for (int x = s.rminx; x < s.rmaxx; ++x) {
                uint64_t key = (uint64_t)(y * gx + x) << 32 | float_bits((float)s.depth);
                kv.emplace_back(key, (uint32_t)i);
            }
    }
    std::stable_sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    keys.resize(kv.size()); vals.resize(kv.size());
    for (size_t k = 0; k < kv.size(); ++k) { keys[k] = kv[k].first; vals[k] = kv[k].second; }
    ranges.assign((size_t)2 * gx * gy, 0);
    for (size_t k = 0; k < kv.size(); ++k) {
        uint32_t t = (uint32_t)(keys[k] >> 32);
        if (k == 0 || t != (uint32_t)(keys[k - 1] >> 32)) ranges[2 * t] = (uint32_t)k;
        if (k + 1 == kv.size() || t != (uint32_t)(keys[k + 1] >> 32)) ranges[2 * t + 1] = (uint32_t)(k + 1);
    }
}

template <class R> inline R rel_margin(R v, R th) { return std::fabs(v - th) / th; }

// Depth compositing rule — one of the three fork semantics SURVEY 8(a) could not settle (the fork's rasteriser is an empty submodule):
//   0 (default): D = sum_i z_i alpha_i T_i                       (what the common depth forks render; no background term)
//   1          : D = sum_i z_i alpha_i T_i / (1 - T_final)       (alpha-normalised expected depth; 0 where nothing was blended)
// A process-global knob of the test oracle (set by oracle.raster_set_depth_mode) so that the long entry-point signatures stay put.
static int g_depth_mode = 0;

template <class R>
void blend_forward(const Problem<R>& pb, const std::vector<Splat<R>>& sp, const std::vector<uint32_t>& vals,
                   const std::vector<uint32_t>& ranges, R* out_color, R* out_depth, R* final_T, uint32_t* n_contrib,
                   int* is_used, R* margin) {
    const int gx = (pb.W + TILE - 1) / TILE;
    const int HW = pb.W * pb.H;
    std::vector<unsigned char> used(pb.P, 0);
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < pb.H; ++py) {
        for (int px = 0; px < pb.W; ++px) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t b = ranges[2 * tile], e = ranges[2 * tile + 1];
            R T = 1, C[3] = {0, 0, 0}, Dz = 0, mg = R(1e30);
            uint32_t contributor = 0, last = 0;
            const R pfx = R(px), pfy = R(py);
            for (uint32_t k = b; k < e; ++k) {
                const Splat<R>& s = sp[vals[k]];
                ++contributor;
                const R dx = s.px - pfx, dy = s.py - pfy;
                const R power = R(-0.5) * (s.ca * dx * dx + s.cc * dy * dy) - s.cb * dx * dy;
                // The margin says how far (relatively) the nearest threshold decision of the pixel is; consumers call a pixel fragile below 1e-5,
                // i.e. they assume two valid fp32 evaluations agree to 1e-5.  Two places where that assumption is not the arithmetic's (round 4,
                // found on a TRAINED map): (i) `power` is a difference of terms that can be 10^3 times larger than itself for a needle-shaped
                // splat (conic eigenvalue ratio 10^3-10^4 after training), so two evaluation orders (FMA contraction on the GPU) differ by
                // ~8 eps32 S with S the sum of the terms' magnitudes — one alpha >= 1/255 test flipped at a relative margin of 1.8e-4; (ii) the
                // transmittance is a product over the entries blended so far and drifts by ~3 ulp per entry (2e-5 after 400).  Both margins are
                // therefore scaled so that "1e-5" keeps meaning "inside the evaluation noise".
                const R S = R(0.5) * (std::fabs(s.ca) * dx * dx + std::fabs(s.cc) * dy * dy) + std::fabs(s.cb * dx * dy);
                const R noise_p = std::max(R(1e-5), R(5e-7) * S);
                if (std::fabs(power) < std::max(R(1e-7), R(5e-7) * S)) mg = 0;  // sign-of-power decision is fragile here
                if (power > R(0)) continue;
                const R a_raw = s.opacity * rexp<R>(power);
                const R alpha = std::min(R(0.99), a_raw);
                mg = std::min(mg, rel_margin<R>(a_raw, R(1) / R(255)) * (R(1e-5) / noise_p));
                if (alpha < R(1) / R(255)) continue;
                const R test_T = T * (R(1) - alpha);
                mg = std::min(mg, rel_margin<R>(test_T, R(0.0001)) * (R(1e-5) / std::max(R(1e-5), R(2e-7) * R(contributor))));
                if (test_T < R(0.0001)) break;
                for (int c = 0; c < 3; ++c) C[c] += s.rgb[c] * alpha * T;
                Dz += s.depth * alpha * T;
                T = test_T;
                last = contributor;
                used[vals[k]] = 1;  // benign race: all writers store 1
            }
            const int pix = py * pb.W + px;
            for (int c = 0; c < 3; ++c) out_color[c * HW + pix] = C[c] + T * pb.bg[c];
            out_depth[pix] = g_depth_mode == 1 ? (T < R(1) ? Dz / (R(1) - T) : R(0)) : Dz;
            final_T[pix] = T;
            n_contrib[pix] = last;
            if (margin) margin[pix] = mg;
        }
    }
    if (is_used) for (int i = 0; i < pb.P; ++i) is_used[i] = used[i];
}

// ---------------------------------------------------------------- backward
template <class R> struct Grads {
    std::vector<double> mean2D, conic, opacity, color, depth;  // per Gaussian: 2,3,1,3,1
};

template <class R>
void blend_backward(const Problem<R>& pb, const std::vector<Splat<R>>& sp, const std::vector<uint32_t>& vals,
                    const std::vector<uint32_t>& ranges, const R* final_T, const uint32_t* n_contrib,
                    const R* dL_dpix /*3HW*/, const R* dL_ddepth /*HW or null*/, const R* out_depth /*HW, forward output*/, Grads<R>& g) {
    const int gx = (pb.W + TILE - 1) / TILE;
    const int HW = pb.W * pb.H;
    g.mean2D.assign((size_t)2 * pb.P, 0); g.conic.assign((size_t)3 * pb.P, 0);
    g.opacity.assign(pb.P, 0); g.color.assign((size_t)3 * pb.P, 0); g.depth.assign(pb.P, 0);
    const R ddelx_dx = R(0.5) * R(pb.W), ddely_dy = R(0.5) * R(pb.H);
    for (int py = 0; py < pb.H; ++py)
        for (int px = 0; px < pb.W; ++px) {
            const int pix = py * pb.W + px;
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t b = ranges[2 * tile];
            const uint32_t last = n_contrib[pix];
            const R T_final = final_T[pix];
            R T = T_final;
            R dpix[4] = {dL_dpix[pix], dL_dpix[HW + pix], dL_dpix[2 * HW + pix], dL_ddepth ? dL_ddepth[pix] : R(0)};
            R accum[4] = {0, 0, 0, 0}, last_c[4] = {0, 0, 0, 0}, last_alpha = 0;
            const R pfx = R(px), pfy = R(py);
            R bg_dot = 0;
            for (int c = 0; c < 3; ++c) bg_dot += pb.bg[c] * dpix[c];
            if (g_depth_mode == 1) {
                // D = N / A, A = 1 - T_final: dD/dalpha_i = (dN/dalpha_i) / A - (N / A^2) T_final / (1 - alpha_i).  The first term is the
                // un-normalised rule with the incoming gradient divided by A; the second has the form of the background term.
                const R A = R(1) - T_final;
                if (A > R(0)) { bg_dot += dpix[3] * out_depth[pix] / A; dpix[3] = dpix[3] / A; } else dpix[3] = 0;
            }
            for (uint32_t kk = last; kk-- > 0;) {
                const uint32_t id = vals[b + kk];
                const Splat<R>& s = sp[id];
                const R dx = s.px - pfx, dy = s.py - pfy;
                const R power = R(-0.5) * (s.ca * dx * dx + s.cc * dy * dy) - s.cb * dx * dy;
                if (power > R(0)) continue;
                const R G = rexp<R>(power);
                const R alpha = std::min(R(0.99), s.opacity * G);
                if (alpha < R(1) / R(255)) continue;
                T = T / (R(1) - alpha);
                const R w = alpha * T;
                R dL_dalpha = 0;
                const R cur[4] = {s.rgb[0], s.rgb[1], s.rgb[2], s.depth};
                for (int c = 0; c < 4; ++c) {
                    accum[c] = last_alpha * last_c[c] + (R(1) - last_alpha) * accum[c];
                    last_c[c] = cur[c];
                    dL_dalpha += (cur[c] - accum[c]) * dpix[c];
                }
                for (int c = 0; c < 3; ++c) g.color[3 * id + c] += (double)(w * dpix[c]);
                g.depth[id] += (double)(w * dpix[3]);
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot;
                const R dL_dG = s.opacity * dL_dalpha;
                const R gdx = G * dx, gdy = G * dy;
                const R dG_ddelx = -gdx * s.ca - gdy * s.cb;
                const R dG_ddely = -gdy * s.cc - gdx * s.cb;
                g.mean2D[2 * id + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                g.mean2D[2 * id + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                g.conic[3 * id + 0] += (double)(R(-0.5) * gdx * dx * dL_dG);
                g.conic[3 * id + 1] += (double)(-gdx * dy * dL_dG);  // true d/d(cb): off-diagonal counted once
                g.conic[3 * id + 2] += (double)(R(-0.5) * gdy * dy * dL_dG);
                g.opacity[id] += (double)(G * dL_dalpha);
            }
        }
}

template <class R>
void sh_backward(int deg, int M, const R* mean, const R* campos, const R* sh, const bool* clamped, const R* dL_drgb_in,
                 R* dL_dsh /* M x 3, written */, R* dL_dmean /* += */) {
    R dL_dRGB[3];
    for (int c = 0; c < 3; ++c) dL_dRGB[c] = clamped[c] ? R(0) : dL_drgb_in[c];
    for (int k = 0; k < 3 * M; ++k) dL_dsh[k] = 0;
    for (int c = 0; c < 3; ++c) dL_dsh[c] = R(SH_C0) * dL_dRGB[c];
    if (deg == 0) return;
    R dir[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    const R len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const R x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    R dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};  // dRGB/d{x,y,z} per channel
    for (int c = 0; c < 3; ++c) {
        dL_dsh[3 + c] = -R(SH_C1) * y * dL_dRGB[c];
        dL_dsh[6 + c] = R(SH_C1) * z * dL_dRGB[c];
        dL_dsh[9 + c] = -R(SH_C1) * x * dL_dRGB[c];
        dx[c] = -R(SH_C1) * sh[9 + c];
        dy[c] = -R(SH_C1) * sh[3 + c];
        dz[c] = R(SH_C1) * sh[6 + c];
    }
    if (deg > 1) {
        const R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        for (int c = 0; c < 3; ++c) {
            dL_dsh[12 + c] = R(SH_C2[0]) * xy * dL_dRGB[c];
            dL_dsh[15 + c] = R(SH_C2[1]) * yz * dL_dRGB[c];
            dL_dsh[18 + c] = R(SH_C2[2]) * (R(2) * zz - xx - yy) * dL_dRGB[c];
            dL_dsh[21 + c] = R(SH_C2[3]) * xz * dL_dRGB[c];
            dL_dsh[24 + c] = R(SH_C2[4]) * (xx - yy) * dL_dRGB[c];
            dx[c] += R(SH_C2[0]) * y * sh[12 + c] + R(SH_C2[2]) * R(2) * -x * sh[18 + c] + R(SH_C2[3]) * z * sh[21 + c] +
                     R(SH_C2[4]) * R(2) * x * sh[24 + c];
            dy[c] += R(SH_C2[0]) * x * sh[12 + c] + R(SH_C2[1]) * z * sh[15 + c] + R(SH_C2[2]) * R(2) * -y * sh[18 + c] +
                     R(SH_C2[4]) * R(2) * -y * sh[24 + c];
            dz[c] += R(SH_C2[1]) * y * sh[15 + c] + R(SH_C2[2]) * R(2) * R(2) * z * sh[18 + c] + R(SH_C2[3]) * x * sh[21 + c];
        }
        if (deg > 2) {
            for (int c = 0; c < 3; ++c) {
                dL_dsh[27 + c] = R(SH_C3[0]) * y * (R(3) * xx - yy) * dL_dRGB[c];
                dL_dsh[30 + c] = R(SH_C3[1]) * xy * z * dL_dRGB[c];
                dL_dsh[33 + c] = R(SH_C3[2]) * y * (R(4) * zz - xx - yy) * dL_dRGB[c];
                dL_dsh[36 + c] = R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy) * dL_dRGB[c];
                dL_dsh[39 + c] = R(SH_C3[4]) * x * (R(4) * zz - xx - yy) * dL_dRGB[c];
                dL_dsh[42 + c] = R(SH_C3[5]) * z * (xx - yy) * dL_dRGB[c];
                dL_dsh[45 + c] = R(SH_C3[6]) * x * (xx - R(3) * yy) * dL_dRGB[c];
                dx[c] += R(SH_C3[0]) * sh[27 + c] * R(3) * R(2) * xy + R(SH_C3[1]) * sh[30 + c] * yz +
                         R(SH_C3[2]) * sh[33 + c] * -R(2) * xy + R(SH_C3[3]) * sh[36 + c] * -R(3) * R(2) * xz +
                         R(SH_C3[4]) * sh[39 + c] * (-R(3) * xx + R(4) * zz - yy) + R(SH_C3[5]) * sh[42 + c] * R(2) * xz +
                         R(SH_C3[6]) * sh[45 + c] * R(3) * (xx - yy);
                dy[c] += R(SH_C3[0]) * sh[27 + c] * R(3) * (xx - yy) + R(SH_C3[1]) * sh[30 + c] * xz +
                         R(SH_C3[2]) * sh[33 + c] * (-R(3) * yy + R(4) * zz - xx) + R(SH_C3[3]) * sh[36 + c] * -R(3) * R(2) * yz +
                         R(SH_C3[4]) * sh[39 + c] * -R(2) * xy + R(SH_C3[5]) * sh[42 + c] * -R(2) * yz +
                         R(SH_C3[6]) * sh[45 + c] * -R(3) * R(2) * xy;
                dz[c] += R(SH_C3[1]) * sh[30 + c] * xy + R(SH_C3[2]) * sh[33 + c] * R(4) * R(2) * yz +
                         R(SH_C3[3]) * sh[36 + c] * R(3) * (R(2) * zz - xx - yy) + R(SH_C3[4]) * sh[39 + c] * R(4) * R(2) * xz +
                         R(SH_C3[5]) * sh[42 + c] * (xx - yy);
            }
        }
    }
    (void)M;
    R dL_ddir[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c) {
        dL_ddir[0] += dx[c] * dL_dRGB[c];
        dL_ddir[1] += dy[c] * dL_dRGB[c];
        dL_ddir[2] += dz[c] * dL_dRGB[c];
    }
    // d(normalised dir)/d(dir)
    const R sum2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    const R invsum32 = R(1) / std::sqrt(sum2 * sum2 * sum2);
    dL_dmean[0] += ((sum2 - dir[0] * dir[0]) * dL_ddir[0] - dir[1] * dir[0] * dL_ddir[1] - dir[2] * dir[0] * dL_ddir[2]) * invsum32;
    dL_dmean[1] += (-dir[0] * dir[1] * dL_ddir[0] + (sum2 - dir[1] * dir[1]) * dL_ddir[1] - dir[2] * dir[1] * dL_ddir[2]) * invsum32;
    dL_dmean[2] += (-dir[0] * dir[2] * dL_ddir[0] - dir[1] * dir[2] * dL_ddir[1] + (sum2 - dir[2] * dir[2]) * dL_ddir[2]) * invsum32;
}

// per-Gaussian chain rule: (mean2D, conic, colour, depth) -> (mean3D, cov3D, sh, scale, rot)
template <class R>
void preprocess_backward(const Problem<R>& pb, const std::vector<Splat<R>>& sp, const Grads<R>& g, R* dL_dmeans3D, R* dL_dcov3D,
                         R* dL_dsh, R* dL_dscales, R* dL_drots) {
    const R fx = R(pb.W) / (R(2) * pb.tanfovx), fy = R(pb.H) / (R(2) * pb.tanfovy);
    for (int i = 0; i < pb.P; ++i) {
        R* dm = dL_dmeans3D + 3 * i;
        dm[0] = dm[1] = dm[2] = 0;
        R dcov[6] = {0, 0, 0, 0, 0, 0};
        if (dL_dsh) for (int k = 0; k < 3 * pb.M; ++k) dL_dsh[(size_t)3 * pb.M * i + k] = 0;
        if (dL_dscales) { dL_dscales[3 * i] = dL_dscales[3 * i + 1] = dL_dscales[3 * i + 2] = 0; }
        if (dL_drots) { for (int k = 0; k < 4; ++k) dL_drots[4 * i + k] = 0; }
        if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0;
        const Splat<R>& s = sp[i];
        if (!s.visible) continue;
        const R* p = pb.means3D + 3 * i;
        // ---- conic -> cov2D -> (cov3D, t)
        R pv[3], Mm[6], tcl[3], abc[3];
        bool clx, cly;
        xform43(p, pb.view, pv);
        ewa_M<R>(pv, fx, fy, pb.tanfovx, pb.tanfovy, pb.view, Mm, tcl, &clx, &cly);
        cov2_from_M(Mm, s.cov3, abc);
        const R a = abc[0], b = abc[1], c = abc[2];
        const R det = a * c - b * b;
        const R gA = (R)g.conic[3 * i], gB = (R)g.conic[3 * i + 1], gC = (R)g.conic[3 * i + 2];
        const R d2inv = R(1) / (det * det + R(0.0000001));
        R dL_da = 0, dL_db = 0, dL_dc = 0;
        if (d2inv != R(0)) {
            dL_da = d2inv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
            dL_dc = d2inv * (-a * a * gC + a * b * gB + (det - a * c) * gA);
            dL_db = d2inv * (R(2) * b * c * gA - (det + R(2) * b * b) * gB + R(2) * a * b * gC);
        }
        // Sigma2 = M S M^T: G2 = [[da, db/2],[db/2, dc]]
        const R h = R(0.5) * dL_db;
        // dL/dSigma3 (full) = M^T G2 M ; unique-variable gradient doubles the off-diagonals
        R GM[6];  // G2 * M (2x3)
        for (int k = 0; k < 3; ++k) { GM[k] = dL_da * Mm[k] + h * Mm[3 + k]; GM[3 + k] = h * Mm[k] + dL_dc * Mm[3 + k]; }
        auto full = [&](int r, int q) { return Mm[r] * GM[q] + Mm[3 + r] * GM[3 + q]; };
        dcov[0] = full(0, 0); dcov[3] = full(1, 1); dcov[5] = full(2, 2);
        dcov[1] = R(2) * full(0, 1); dcov[2] = R(2) * full(0, 2); dcov[4] = R(2) * full(1, 2);
        // dL/dM = 2 * G2 * M * Sigma3
        const R S[9] = {s.cov3[0], s.cov3[1], s.cov3[2], s.cov3[1], s.cov3[3], s.cov3[4], s.cov3[2], s.cov3[4], s.cov3[5]};
        R dM[6];
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 3; ++k)
                dM[3 * r + k] = R(2) * (GM[3 * r] * S[k] + GM[3 * r + 1] * S[3 + k] + GM[3 * r + 2] * S[6 + k]);
        // M = J * Rw2c  ->  dL/dJ = dL/dM * Rw2c^T ; Rw2c[r][k] = view[4k + r]
        auto dJ = [&](int r, int col) { return dM[3 * r] * pb.view[col] + dM[3 * r + 1] * pb.view[4 + col] + dM[3 * r + 2] * pb.view[8 + col]; };
        const R dJ00 = dJ(0, 0), dJ02 = dJ(0, 2), dJ11 = dJ(1, 1), dJ12 = dJ(1, 2);
        const R tz = R(1) / tcl[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const R xm = clx ? R(0) : R(1), ym = cly ? R(0) : R(1);
        const R dtx = xm * (-fx * tz2 * dJ02);
        const R dty = ym * (-fy * tz2 * dJ12);
        const R dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (R(2) * fx * tcl[0]) * tz3 * dJ02 + (R(2) * fy * tcl[1]) * tz3 * dJ12;
        // t = Rw2c p + trans -> dL/dp = Rw2c^T dL/dt
        for (int k = 0; k < 3; ++k) dm[k] += pb.view[4 * k] * dtx + pb.view[4 * k + 1] * dty + pb.view[4 * k + 2] * dtz;
        // ---- depth -> mean (view-space z = row 2 of Rw2c . p + tz)
        const R gd = (R)g.depth[i];
        for (int k = 0; k < 3; ++k) dm[k] += pb.view[4 * k + 2] * gd;
        // ---- mean2D (already scaled to NDC units) -> mean3D through the projection
        {
            R ph[4];
            xform44(p, pb.proj, ph);
            const R mw = R(1) / (ph[3] + R(0.0000001));
            const R mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
            const R g0 = (R)g.mean2D[2 * i], g1 = (R)g.mean2D[2 * i + 1];
            for (int k = 0; k < 3; ++k)
                dm[k] += (pb.proj[4 * k] * mw - pb.proj[4 * k + 3] * mul1) * g0 + (pb.proj[4 * k + 1] * mw - pb.proj[4 * k + 3] * mul2) * g1;
        }
        // ---- colour -> SH (and mean through the view direction)
        if (!pb.colors_precomp && dL_dsh) {
            R dcol[3] = {(R)g.color[3 * i], (R)g.color[3 * i + 1], (R)g.color[3 * i + 2]};
            sh_backward<R>(pb.D, pb.M, p, pb.campos, pb.shs + (size_t)3 * pb.M * i, s.clamped, dcol, dL_dsh + (size_t)3 * pb.M * i, dm);
        }
        // ---- cov3D -> scale, rotation
        if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dcov[k];
        if (!pb.cov3D_precomp && dL_dscales && dL_drots) {
            const R* q = pb.rotations + 4 * i;
            const R* sc = pb.scales + 3 * i;
            R Rm[9];
            quat_to_R(q, Rm);
            const R sv[3] = {pb.scale_modifier * sc[0], pb.scale_modifier * sc[1], pb.scale_modifier * sc[2]};
            // G3 full symmetric (off-diagonals halved back)
            const R G3[9] = {dcov[0], R(0.5) * dcov[1], R(0.5) * dcov[2], R(0.5) * dcov[1], dcov[3], R(0.5) * dcov[4],
                             R(0.5) * dcov[2], R(0.5) * dcov[4], dcov[5]};
            // Sigma = L L^T, L = Rot S ; dL/dL = 2 G3 L
            R dLm[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k)
                    dLm[3 * r + k] = R(2) * (G3[3 * r] * Rm[k] + G3[3 * r + 1] * Rm[3 + k] + G3[3 * r + 2] * Rm[6 + k]) * sv[k];
            R dR[9];
            for (int k = 0; k < 3; ++k) {
                R acc = 0;
                for (int r = 0; r < 3; ++r) { acc += dLm[3 * r + k] * Rm[3 * r + k]; dR[3 * r + k] = dLm[3 * r + k] * sv[k]; }
                dL_dscales[3 * i + k] = pb.scale_modifier * acc;
            }
            const R x = q[0], y = q[1], z = q[2], r = q[3];
            // dR[row*3+col]
            const R d00 = dR[0], d01 = dR[1], d02 = dR[2], d10 = dR[3], d11 = dR[4], d12 = dR[5], d20 = dR[6], d21 = dR[7], d22 = dR[8];
            dL_drots[4 * i + 0] = R(2) * (y * (d01 + d10) + z * (d02 + d20) + r * (d21 - d12)) - R(4) * x * (d11 + d22);
            dL_drots[4 * i + 1] = R(2) * (x * (d01 + d10) + z * (d12 + d21) + r * (d02 - d20)) - R(4) * y * (d00 + d22);
            dL_drots[4 * i + 2] = R(2) * (x * (d02 + d20) + y * (d12 + d21) + r * (d10 - d01)) - R(4) * z * (d00 + d11);
            dL_drots[4 * i + 3] = R(2) * (x * (d21 - d12) + y * (d02 - d20) + z * (d10 - d01));
        }
    }
}

template <class R>
int forward_impl(int P, int D, int M, const R* bg, int W, int H, const R* means3D, const R* shs, const R* colors_precomp,
                 const R* opacities, const R* scales, R scale_modifier, const R* rotations, const R* cov3D_precomp,
                 const R* view, const R* proj, const R* campos, R tanfovx, R tanfovy, int prefiltered, R* out_color,
                 R* out_depth, int* radii, int* is_used, R* geom /* P x 12: px,py,depth,ca,cb,cc,opacity,r,g,b,tiles,visible */,
                 uint64_t* keys_out, uint32_t* vals_out, long long cap, uint32_t* ranges_out, R* final_T, uint32_t* n_contrib,
                 R* margin) {
    Problem<R> pb{P, D, M, W, H, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                  scale_modifier, view, proj, campos, tanfovx, tanfovy, prefiltered};
    std::vector<Splat<R>> sp;
    preprocess(pb, sp);
    std::vector<uint64_t> keys; std::vector<uint32_t> vals, ranges;
    bin_and_sort(pb, sp, keys, vals, ranges);
    const int HW = W * H;
    std::vector<R> fT(HW); std::vector<uint32_t> nc(HW);
    blend_forward(pb, sp, vals, ranges, out_color, out_depth, fT.data(), nc.data(), is_used, margin);
    for (int i = 0; i < P; ++i) {
        if (radii) radii[i] = sp[i].visible ? sp[i].radius : 0;
        if (geom) {
            R* gr = geom + 12 * i;
            const Splat<R>& s = sp[i];
            gr[0] = s.px; gr[1] = s.py; gr[2] = s.depth; gr[3] = s.ca; gr[4] = s.cb; gr[5] = s.cc; gr[6] = s.opacity;
            gr[7] = s.rgb[0]; gr[8] = s.rgb[1]; gr[9] = s.rgb[2];
            gr[10] = s.visible ? R((s.rmaxx - s.rminx) * (s.rmaxy - s.rminy)) : R(0);
            gr[11] = s.visible ? R(1) : R(0);
        }
    }
    if (final_T) std::copy(fT.begin(), fT.end(), final_T);
    if (n_contrib) std::copy(nc.begin(), nc.end(), n_contrib);
    if (ranges_out) std::copy(ranges.begin(), ranges.end(), ranges_out);
    const long long n = (long long)vals.size();
    if (keys_out && vals_out) {
        const long long m = std::min(n, cap);
        std::copy(keys.begin(), keys.begin() + m, keys_out);
        std::copy(vals.begin(), vals.begin() + m, vals_out);
    }
    return (int)n;
}

template <class R>
int backward_impl(int P, int D, int M, const R* bg, int W, int H, const R* means3D, const R* shs, const R* colors_precomp,
                  const R* opacities, const R* scales, R scale_modifier, const R* rotations, const R* cov3D_precomp,
                  const R* view, const R* proj, const R* campos, R tanfovx, R tanfovy, const R* dL_dpix, const R* dL_ddepth,
                  R* dL_dmeans2D /*P x 3*/, R* dL_dconic /*P x 3*/, R* dL_dopacity, R* dL_dcolors, R* dL_ddepths,
                  R* dL_dmeans3D, R* dL_dcov3D, R* dL_dsh, R* dL_dscales, R* dL_drots) {
    Problem<R> pb{P, D, M, W, H, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                  scale_modifier, view, proj, campos, tanfovx, tanfovy, 0};
    std::vector<Splat<R>> sp;
    preprocess(pb, sp);
    std::vector<uint64_t> keys; std::vector<uint32_t> vals, ranges;
    bin_and_sort(pb, sp, keys, vals, ranges);
    const int HW = W * H;
    std::vector<R> color(3 * (size_t)HW), depth(HW), fT(HW); std::vector<uint32_t> nc(HW);
    blend_forward(pb, sp, vals, ranges, color.data(), depth.data(), fT.data(), nc.data(), (int*)nullptr, (R*)nullptr);
    Grads<R> g;
    blend_backward(pb, sp, vals, ranges, fT.data(), nc.data(), dL_dpix, dL_ddepth, depth.data(), g);
    for (int i = 0; i < P; ++i) {
        if (dL_dmeans2D) { dL_dmeans2D[3 * i] = (R)g.mean2D[2 * i]; dL_dmeans2D[3 * i + 1] = (R)g.mean2D[2 * i + 1]; dL_dmeans2D[3 * i + 2] = 0; }
        if (dL_dconic) for (int k = 0; k < 3; ++k) dL_dconic[3 * i + k] = (R)g.conic[3 * i + k];
        if (dL_dopacity) dL_dopacity[i] = (R)g.opacity[i];
        if (dL_dcolors) for (int k = 0; k < 3; ++k) dL_dcolors[3 * i + k] = (R)g.color[3 * i + k];
        if (dL_ddepths) dL_ddepths[i] = (R)g.depth[i];
    }
    preprocess_backward(pb, sp, g, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drots);
    return (int)vals.size();
}

}  // namespace

extern "C" {
void oracle_raster_set_depth_mode(int m) { g_depth_mode = m; }

#define FWD_ARGS(R)                                                                                                        \
    int P, int D, int M, const R *bg, int W, int H, const R *means3D, const R *shs, const R *colors_precomp,               \
        const R *opacities, const R *scales, R scale_modifier, const R *rotations, const R *cov3D_precomp, const R *view,  \
        const R *proj, const R *campos, R tanfovx, R tanfovy, int prefiltered, R *out_color, R *out_depth, int *radii,     \
        int *is_used, R *geom, uint64_t *keys_out, uint32_t *vals_out, long long cap, uint32_t *ranges_out, R *final_T,    \
        uint32_t *n_contrib, R *margin
#define FWD_PASS                                                                                                           \
    P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, view,    \
        proj, campos, tanfovx, tanfovy, prefiltered, out_color, out_depth, radii, is_used, geom, keys_out, vals_out, cap,  \
        ranges_out, final_T, n_contrib, margin

int oracle_raster_forward_f32(FWD_ARGS(float)) { return forward_impl<float>(FWD_PASS); }
int oracle_raster_forward_f64(FWD_ARGS(double)) { return forward_impl<double>(FWD_PASS); }

#define BWD_ARGS(R)                                                                                                        \
    int P, int D, int M, const R *bg, int W, int H, const R *means3D, const R *shs, const R *colors_precomp,               \
        const R *opacities, const R *scales, R scale_modifier, const R *rotations, const R *cov3D_precomp, const R *view,  \
        const R *proj, const R *campos, R tanfovx, R tanfovy, const R *dL_dpix, const R *dL_ddepth, R *dL_dmeans2D,        \
        R *dL_dconic, R *dL_dopacity, R *dL_dcolors, R *dL_ddepths, R *dL_dmeans3D, R *dL_dcov3D, R *dL_dsh,               \
        R *dL_dscales, R *dL_drots
#define BWD_PASS                                                                                                           \
    P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, view,    \
        proj, campos, tanfovx, tanfovy, dL_dpix, dL_ddepth, dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors, dL_ddepths,   \
        dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drots

int oracle_raster_backward_f32(BWD_ARGS(float)) { return backward_impl<float>(BWD_PASS); }
int oracle_raster_backward_f64(BWD_ARGS(double)) { return backward_impl<double>(BWD_PASS); }

}  // extern "C"
