// Tracker front-end on the device: what Tracker.downsample_and_make_pointcloud2 does with torch-CPU and
// numpy ops on every frame [REF mp_Tracker.py:415-431] — pick the pre-selected pixels, convert depth, drop the zero-depth
// picks (order preserved), back-project with the precomputed (u-cx)/fx, (v-cy)/fy, scale the colours, and list the picks
// within depth_trunc — as ONE single-workgroup launch whose outputs feed the tracker's device-pointer overloads directly.
//
// One workgroup is the right size: a frame has 8-12 k picks, the work is two order-preserving compactions (running offsets
// across 1024-wide chunks), and a multi-workgroup scan would cost more launches than the whole job takes.
#include <cstdint>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/gsicp_hip.h"

namespace gsicp {
extern thread_local std::string g_last_error;
namespace {

template <class DepthT>
__global__ __launch_bounds__(1024) void make_pointcloud_kernel(int n_pick, const long long* __restrict__ pick_idx, const float* __restrict__ x_pre,
                                                               const float* __restrict__ y_pre, const DepthT* __restrict__ depth,
                                                               const unsigned char* __restrict__ rgb, float depth_scale, float depth_trunc,
                                                               float* __restrict__ points, float* __restrict__ colors, float* __restrict__ z_values,
                                                               int* __restrict__ trackable_idx, int* __restrict__ counts) {
    __shared__ unsigned s_keep[16], s_trk[16];
    __shared__ unsigned s_base_keep, s_base_trk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_base_keep = 0; s_base_trk = 0; }
    __syncthreads();
    for (int base = 0; base < n_pick; base += 1024) {
        const int i = base + tid;
        float z = 0.f;
        long long pix = 0;
        if (i < n_pick) {
            pix = pick_idx[i];
            z = (float)depth[pix] / depth_scale;            // depth_img.astype(np.float32) ... / self.depth_scale  [REF :418]
        }
        const bool keep = i < n_pick && z != 0.f;            // zero_filter = torch.where(z_values != 0)             [REF :419]
        const bool trk = keep && z <= depth_trunc;           // filter = torch.where(z_values[zero_filter] <= trunc)  [REF :420]
        const unsigned long long bk = __ballot(keep), bt = __ballot(trk);
        if (lane == 0) { s_keep[wave] = (unsigned)__popcll(bk); s_trk[wave] = (unsigned)__popcll(bt); }
        __syncthreads();
        unsigned pk = s_base_keep + (unsigned)__popcll(bk & ((1ull << lane) - 1ull));
        unsigned pt = s_base_trk + (unsigned)__popcll(bt & ((1ull << lane) - 1ull));
        unsigned tk = 0, tt = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { pk += s_keep[w]; pt += s_trk[w]; }
            tk += s_keep[w]; tt += s_trk[w];
        }
        if (keep) {
            points[3 * (size_t)pk] = x_pre[i] * z;           // x = self.x_pre[zero_filter] * z_values                [REF :424-426]
            points[3 * (size_t)pk + 1] = y_pre[i] * z;
            points[3 * (size_t)pk + 2] = z;
            z_values[pk] = z;
            if (rgb && colors) {                             // colors = rgb.reshape(-1,3).float()[idx] / 255          [REF :417, 427]
#pragma unroll
                for (int c = 0; c < 3; ++c) colors[3 * (size_t)pk + c] = (float)rgb[3 * pix + c] / 255.f;
            }
            if (trk) trackable_idx[pt] = (int)pk;            // indices INTO the compacted arrays, ascending
        }
        __syncthreads();
        if (tid == 0) { s_base_keep += tk; s_base_trk += tt; }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = (int)s_base_keep; counts[1] = (int)s_base_trk; }
}

}  // namespace
}  // namespace gsicp

using namespace gsicp;

extern "C" int gsicp_frontend_make_pointcloud(int n_pick, const long long* pick_idx, const float* x_pre, const float* y_pre, const void* depth,
                                              int depth_type, const unsigned char* rgb, float depth_scale, float depth_trunc, float* points,
                                              float* colors, float* z_values, int* trackable_idx, int* counts, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (n_pick < 0 || (n_pick > 0 && (!pick_idx || !x_pre || !y_pre || !depth || !points || !z_values || !trackable_idx)) || !counts) {
        g_last_error = "gsicp_frontend_make_pointcloud: bad arguments"; return -2;
    }
    if (depth_scale == 0.f) { g_last_error = "gsicp_frontend_make_pointcloud: depth_scale is zero"; return -2; }
    switch (depth_type) {
    case 0:
        hipLaunchKernelGGL(make_pointcloud_kernel<uint16_t>, dim3(1), dim3(1024), 0, stream, n_pick, pick_idx, x_pre, y_pre, (const uint16_t*)depth,
                           rgb, depth_scale, depth_trunc, points, colors, z_values, trackable_idx, counts);
        break;
    case 1:
        hipLaunchKernelGGL(make_pointcloud_kernel<float>, dim3(1), dim3(1024), 0, stream, n_pick, pick_idx, x_pre, y_pre, (const float*)depth, rgb,
                           depth_scale, depth_trunc, points, colors, z_values, trackable_idx, counts);
        break;
    default:
        g_last_error = "gsicp_frontend_make_pointcloud: depth_type must be 0 (uint16) or 1 (float32)"; return -2;
    }
    if (hipGetLastError() != hipSuccess) { g_last_error = "gsicp_frontend_make_pointcloud: kernel launch failed"; return -1; }
    return 0;
}
