// stream_ceiling.hip -- what a read-modify-write stream over 8-byte voxels can reach on this GPU: the ceiling the whole-map kernels
// (k_decay: every TSDF voxel read, its weight scaled, written back) are measured against, beside the 8 TB/s paper peak.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>

// (a) one 8-byte voxel per lane per iteration, 512-thread workgroups striding over 4 KiB blocks: k_decay's access pattern without its books
__global__ __launch_bounds__(512) void rmw8(float2* p, size_t nblocks, float f) {
  for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) { float2 v = p[b * 512 + threadIdx.x]; v.y *= f; p[b * 512 + threadIdx.x] = v; }
}
// (b) eight blocks in flight per workgroup iteration (k_decay's batching)
__global__ __launch_bounds__(512) void rmw8x8(float2* p, size_t nblocks, float f) {
  for (size_t b = (size_t)blockIdx.x * 8; b < nblocks; b += (size_t)gridDim.x * 8) {
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = b + j < nblocks ? p[(b + j) * 512 + threadIdx.x] : make_float2(0, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) if (b + j < nblocks) { v[j].y *= f; p[(b + j) * 512 + threadIdx.x] = v[j]; }
  }
}
// (c) 16 bytes per lane (two voxels), flat grid-stride: the widest per-lane access
__global__ __launch_bounds__(256) void rmw16(float4* p, size_t n, float f) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.y *= f; v.w *= f; p[i] = v; }
}
// (d) read only / (e) write only, 16 bytes per lane
__global__ __launch_bounds__(256) void rd16(const float4* p, size_t n, float* out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) *out = s;
}
__global__ __launch_bounds__(256) void wr16(float4* p, size_t n, float f) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(f, f, f, f);
}

// (f) one wavefront per block, eight registers per lane (the barrier-free k_decay's pattern), wavefront g takes blocks g, g + stride, ...
__global__ __launch_bounds__(512) void rmw_wave(float2* p, size_t nblocks, float f) {
  const size_t stride = (size_t)gridDim.x * 8;
  const int lane = threadIdx.x & 63;
  for (size_t b = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6); b < nblocks; b += stride) {
    float2* vp = p + b * 512 + lane;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = vp[r * 64];
#pragma unroll
    for (int r = 0; r < 8; r++) { v[r].y *= f; vp[r * 64] = v[r]; }
  }
}
template <typename F> static float time_us(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1000.0f / reps;
}
int main() {
  const size_t nblocks = 146221;                 // the LiDAR map of tools/maintenance_bw.py: 0.6 GB of TSDF
  const size_t bytes = nblocks * 4096;
  float2* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
  float* out; hipMalloc(&out, 4);
  const double gb = (double)bytes / 1e9;
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    const float a = time_us([&] { rmw8<<<grid, 512>>>(p, nblocks, 0.999f); }, 20);
    const float b = time_us([&] { rmw8x8<<<grid, 512>>>(p, nblocks, 0.999f); }, 20);
    const float c = time_us([&] { rmw16<<<grid, 256>>>((float4*)p, bytes / 16, 0.999f); }, 20);
    const float d = time_us([&] { rd16<<<grid, 256>>>((const float4*)p, bytes / 16, out); }, 20);
    const float e = time_us([&] { wr16<<<grid, 256>>>((float4*)p, bytes / 16, 0.5f); }, 20);
    const float w = time_us([&] { rmw_wave<<<grid, 512>>>(p, nblocks, 0.999f); }, 20);
    printf("grid %5d: wave-per-block %.0f us %.2f TB/s | ", grid, w, 2 * gb / w * 1e3);
    printf("rmw 8B/lane %.0f us %.2f TB/s | rmw 8 blocks in flight %.0f us %.2f TB/s | rmw 16B/lane %.0f us %.2f TB/s | read %.0f us %.2f TB/s | write %.0f us %.2f TB/s\n",
           a, 2 * gb / a * 1e3, b, 2 * gb / b * 1e3, c, 2 * gb / c * 1e3, d, gb / d * 1e3, e, gb / e * 1e3);
  }
  return 0;
}
