// pmc_calibrate.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts in the access patterns of this library.
// /opt/skills/guides/MI355X_MICROARCH.md (HBM): "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) --
// double it before comparing with a byte count.  Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute."  tools/summarize_profile.py applies the x2 to every kernel; the paths here move 8-byte voxels (one per lane, a
// 4 KiB block per 512-thread workgroup), 16-byte hash entries at scattered addresses and 64 contiguous bytes per lane (the marking pass's columns).
// One launch per pattern over a buffer far beyond the 256 MiB last-level cache, every kernel name = its pattern; run under
//   rocprofv3 --pmc FETCH_SIZE -- /tmp/pmc_calibrate      and      rocprofv3 --pmc WRITE_SIZE -- /tmp/pmc_calibrate
// (tools/pmc_calibrate.sh does both and prints bytes moved / bytes reported per pattern -> profiles/r06_pmc_calibration.json).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/pmc_calibrate.hip -o /tmp/pmc_calibrate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdint>
#include <cstdio>

// a 4 KiB block per 512-thread workgroup, one 8-byte voxel per lane: the TSDF update's read / the decay's read-modify-write / a block write
__global__ __launch_bounds__(512) void cal_read8_block(const float2* p, size_t nblocks, float* out) {
  float s = 0;
  for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) { const float2 v = p[b * 512 + threadIdx.x]; s += v.x + v.y; }
  if (s == 123.456f) *out = s;
}
__global__ __launch_bounds__(512) void cal_write8_block(float2* p, size_t nblocks, float f) {
  for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) p[b * 512 + threadIdx.x] = make_float2(f, f);
}
__global__ __launch_bounds__(512) void cal_rmw8_block(float2* p, size_t nblocks, float f) {
  for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) { float2 v = p[b * 512 + threadIdx.x]; v.y *= f; p[b * 512 + threadIdx.x] = v; }
}
// 16 bytes per lane, coalesced: the guide's calibrated case (the reference point of this run)
__global__ __launch_bounds__(256) void cal_read16_stream(const float4* p, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) *out = s;
}
// one 16-byte hash entry per lane, every lane in a 128-byte line of its own (a multiplicative permutation of the line index): the probes of the flushes,
// the sphere tracing and the colour / ESDF workers.  Bytes USED: 16 per access; bytes a 64-B / 128-B line costs: 64 / 128.
__global__ __launch_bounds__(256) void cal_read16_scattered(const uint4* p, size_t nlines, size_t naccess, uint32_t* out) {
  uint32_t s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < naccess; i += (size_t)gridDim.x * blockDim.x) {
    const size_t line = (i * 2654435761ull) % nlines;           // (nlines is a power of two times an odd number: distinct lines for distinct i < nlines)
    const uint4 v = p[line * 8 + (i & 7)];                      // 8 entries of 16 B per 128-B line
    s += v.x ^ v.w;
  }
  if (s == 0xDEADBEEFu) *out = s;
}
// 64 contiguous bytes per lane as four 16-byte loads, a wavefront = one 4 KiB block: the ESDF marking pass's column reads
__global__ __launch_bounds__(256) void cal_read64_per_lane(const float4* p, size_t nblocks, float* out) {
  float s = 0;
  const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t b = wave; b < nblocks; b += nwave) {
    const float4* c = p + b * 256 + (size_t)lane * 4;
    const float4 a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    s += a0.x + a1.y + a2.z + a3.w;
  }
  if (s == 123.456f) *out = s;
}
// one 8-byte voxel per lane at a scattered address (the sphere tracing's voxel reads): 8 bytes used per 128-B line touched
__global__ __launch_bounds__(256) void cal_read8_scattered(const float2* p, size_t nlines, size_t naccess, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < naccess; i += (size_t)gridDim.x * blockDim.x) {
    const size_t line = (i * 2654435761ull) % nlines;
    const float2 v = p[line * 16 + (i & 15)];
    s += v.x + v.y;
  }
  if (s == 123.456f) *out = s;
}

int main() {
  const size_t bytes = (size_t)2 << 30;                        // 2 GiB: eight times the last-level cache
  void* buf = nullptr; float* out = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes); hipDeviceSynchronize();
  const size_t nblocks = bytes / 4096, n16 = bytes / 16, nlines = bytes / 128, nacc = nlines / 2;      // scattered: half of the lines, each touched once
  const int g = 256 * 8;
  hipLaunchKernelGGL(cal_read16_stream, dim3(g), dim3(256), 0, 0, (const float4*)buf, n16, out); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_read8_block, dim3(g), dim3(512), 0, 0, (const float2*)buf, nblocks, out); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_read64_per_lane, dim3(g), dim3(256), 0, 0, (const float4*)buf, nblocks, out); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_read16_scattered, dim3(g), dim3(256), 0, 0, (const uint4*)buf, nlines, nacc, (uint32_t*)out); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_read8_scattered, dim3(g), dim3(256), 0, 0, (const float2*)buf, nlines, nacc, out); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_write8_block, dim3(g), dim3(512), 0, 0, (float2*)buf, nblocks, 1.0f); hipDeviceSynchronize();
  hipLaunchKernelGGL(cal_rmw8_block, dim3(g), dim3(512), 0, 0, (float2*)buf, nblocks, 0.5f); hipDeviceSynchronize();
  // what each launch moved, for the script that divides by what the counters report
  std::printf("{\"bytes\": %zu, \"cal_read16_stream\": {\"read\": %zu}, \"cal_read8_block\": {\"read\": %zu}, \"cal_read64_per_lane\": {\"read\": %zu}, "
              "\"cal_read16_scattered\": {\"read_used\": %zu, \"read_lines64\": %zu, \"read_lines128\": %zu}, "
              "\"cal_read8_scattered\": {\"read_used\": %zu, \"read_lines64\": %zu, \"read_lines128\": %zu}, "
              "\"cal_write8_block\": {\"write\": %zu}, \"cal_rmw8_block\": {\"read\": %zu, \"write\": %zu}}\n",
              bytes, bytes, bytes, bytes, nacc * 16, nacc * 64, nacc * 128, nacc * 8, nacc * 64, nacc * 128, bytes, bytes, bytes);
  hipFree(buf); hipFree(out);
  return 0;
}
