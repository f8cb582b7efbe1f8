// EXPERIMENT (recorded in DESIGN.md / profiles/, not on the product path): the batched 3x3 contraction Sigma = R diag(s^2) R^T of
// cov_fromqs_kernel / preprocess_kernel on the matrix cores vs on the vector ALU.  north_star asks that MFMA be used "only for the batched
// 3x3 covariance contractions" and that the choice be evidenced; SURVEY 8(d) predicted no gain (fp32 MFMA runs at the vector rate and a
// 3x3 product fills 9/16 of a 4x4 block).  Both variants read (quaternion xyzw, scale) per Gaussian and write the 6 unique entries.
//   variant 0  VALU : one thread per Gaussian, 3x3 in registers (what the product kernels do)
//   variant 1  MFMA : v_mfma_f32_4x4x1_16B_f32 — 16 independent 4x4 outer-product accumulators per wave; four lanes share a Gaussian,
//                     lane i supplies row i of M = R diag(s) as both A and B operand, k = 0..2 accumulates M M^T; lanes 0..2 of each group
//                     then hold columns 0..2 of Sigma.
#include <hip/hip_runtime.h>

namespace {
__device__ inline void rot_row(const float4 q, int i, float& r0, float& r1, float& r2) {
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    if (i == 0)      { r0 = 1.f - 2.f * (y * y + z * z); r1 = 2.f * (x * y - w * z);       r2 = 2.f * (x * z + w * y); }
    else if (i == 1) { r0 = 2.f * (x * y + w * z);       r1 = 1.f - 2.f * (x * x + z * z); r2 = 2.f * (y * z - w * x); }
    else             { r0 = 2.f * (x * z - w * y);       r1 = 2.f * (y * z + w * x);       r2 = 1.f - 2.f * (x * x + y * y); }
}

__global__ __launch_bounds__(256) void cov3_valu_kernel(int n, const float4* __restrict__ q, const float* __restrict__ s, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 qq = q[i];
    const float s0 = s[3 * (size_t)i], s1 = s[3 * (size_t)i + 1], s2 = s[3 * (size_t)i + 2];
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a, b, c;
        rot_row(qq, r, a, b, c);
        M[3 * r] = a * s0; M[3 * r + 1] = b * s1; M[3 * r + 2] = c * s2;
    }
    float* o = out + 6 * (size_t)i;
    o[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    o[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    o[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    o[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    o[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    o[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void cov3_mfma_kernel(int n, const float4* __restrict__ q, const float* __restrict__ s, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 2;          // Gaussian of this 4-lane group
    const int row = lane & 3;
    const bool live = g < n;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (live && row < 3) {
        const float4 qq = q[g];
        float a, b, c;
        rot_row(qq, row, a, b, c);
        m0 = a * s[3 * (size_t)g]; m1 = b * s[3 * (size_t)g + 1]; m2 = c * s[3 * (size_t)g + 2];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m0, m0, acc, 0, 0, 0);   // D[i][j] += M[i][0] M[j][0], 16 Gaussians at once
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m1, m1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m2, m2, acc, 0, 0, 0);
    if (!live) return;
    float* o = out + 6 * (size_t)g;                                // lane j of the group holds column j: acc[i] = Sigma[i][j]
    if (row == 0) { o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; }
    else if (row == 1) { o[3] = acc[1]; o[4] = acc[2]; }
    else if (row == 2) { o[5] = acc[2]; }
}
}  // namespace

extern "C" int gsicp_exp_cov3(int n, const float* quats_xyzw, const float* scales, float* out6, int variant, int iters, float* us_per_launch,
                              void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    auto launch = [&] {
        if (variant == 0) hipLaunchKernelGGL(cov3_valu_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, (const float4*)quats_xyzw, scales, out6);
        else hipLaunchKernelGGL(cov3_mfma_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, stream, n, (const float4*)quats_xyzw, scales, out6);
    };
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0, stream);
    for (int i = 0; i < iters; ++i) launch();
    (void)hipEventRecord(e1, stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (us_per_launch) *us_per_launch = 1e3f * ms / (float)(iters > 0 ? iters : 1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
