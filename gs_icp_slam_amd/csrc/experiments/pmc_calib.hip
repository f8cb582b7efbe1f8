// CALIBRATION of rocprofv3's FETCH_SIZE / WRITE_SIZE on this repo's own access patterns (MI355X_MICROARCH.md: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Four kernels of KNOWN compulsory traffic, each launched a few times by tools/pmc_calibration.py
// under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
//   calib_stream_read   : N float4 read once, coalesced (16 B / lane)                       -> 16 N bytes fetched
//   calib_gather48      : N 48-byte records (3 x float4, the blend kernels' SplatRec) read ONCE each in a random permutation, all lanes
//                         gathering in parallel like the blend kernels' staging step         -> 48 N bytes fetched (+ 4 N index bytes)
//   calib_stream_write  : N float4 written once, coalesced                                  -> 16 N bytes written
//   calib_record_write48: N 48-byte records written once each in a random permutation (the backward's entry records) -> 48 N bytes written
// The buffers are sized past the 4 MiB L2 but the counters sit on the L2's memory side, so Infinity-Cache hits are counted anyway.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
__global__ __launch_bounds__(256) void calib_stream_read(size_t n, const float4* __restrict__ src, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather48(size_t n, const uint32_t* __restrict__ perm, const float4* __restrict__ rec, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4* r = rec + 3 * (size_t)perm[i];
        const float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_stream_write(size_t n, float4* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ __launch_bounds__(256) void calib_record_write48(size_t n, const uint32_t* __restrict__ perm, float4* __restrict__ rec) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4* r = rec + 3 * (size_t)perm[i];
        r[0] = make_float4((float)i, 0.f, 0.f, 0.f); r[1] = make_float4(1.f, 1.f, 1.f, 1.f); r[2] = make_float4(2.f, 2.f, 2.f, 2.f);
    }
}
}  // namespace

extern "C" int gsicp_exp_pmc_calib(int which, size_t n, const void* perm, void* buf, void* sink, int launches, void* stream_v) {
    hipStream_t s = (hipStream_t)stream_v;
    const dim3 grid(2048), block(256);
    for (int l = 0; l < launches; ++l) {
        if (which == 0) hipLaunchKernelGGL(calib_stream_read, grid, block, 0, s, n, (const float4*)buf, (float*)sink);
        else if (which == 1) hipLaunchKernelGGL(calib_gather48, grid, block, 0, s, n, (const uint32_t*)perm, (const float4*)buf, (float*)sink);
        else if (which == 2) hipLaunchKernelGGL(calib_stream_write, grid, block, 0, s, n, (float4*)buf);
        else hipLaunchKernelGGL(calib_record_write48, grid, block, 0, s, n, (const uint32_t*)perm, (float4*)buf);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
