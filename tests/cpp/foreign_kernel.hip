// A kernel that is NOT part of libnvblox_hip reads the map in place through include/nvblox_hip_device.h -- the role
// esdf_and_gradients_conversions.cu:88-125 (GPULayerView + gpu_indexing.cuh) plays in the reference node.  Builds a small
// map through the C-ABI, samples the ESDF and the TSDF on a dense voxel grid from its own kernel, and prints sums that
// tests/test_cpp_facade.py compares with the library's own nvbx_esdf_dense_grid / nvbx_get_blocks.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "nvblox_hip.h"
#include "nvblox_hip_device.h"

#define CHECK(x) do { if ((x) != 0) { std::fprintf(stderr, "failed: %s (%s)\n", #x, nvbx_last_error()); return 1; } } while (0)

__global__ void k_sample(nvbx_device_view v, int3 mn, int3 sz, float unknown, float* esdf_out, float* tsdf_w_out) {
  const int64_t n = (int64_t)sz.x * sz.y * sz.z;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = (int)(i % sz.z), y = (int)((i / sz.z) % sz.y), x = (int)(i / ((int64_t)sz.z * sz.y));     // x*(Ny*Nz) + y*Nz + z
    const int gx = mn.x + x, gy = mn.y + y, gz = mn.z + z;
    esdf_out[i] = nvbx_dev_esdf_distance_m(v, gx, gy, gz, unknown);
    const uint32_t s = nvbx_dev_find_block(v, gx >> 3, gy >> 3, gz >> 3, NVBX_LAYER_TSDF);
    tsdf_w_out[i] = nvbx_dev_slot_ok(s) ? nvbx_dev_tsdf_voxel(v, s, gx & 7, gy & 7, gz & 7).weight : 0.0f;
  }
}

int main() {
  hipStream_t stream; if (hipStreamCreate(&stream) != hipSuccess) return 1;
  nvbx_mapper_params p; nvbx_default_params(&p);
  nvbx_mapper* m = nullptr;
  CHECK(nvbx_mapper_create(0, stream, &p, 1 << 13, &m));
  // a wall 2 m in front of a 160x120 camera at the origin looking along +x (camera z = world x)
  const int rows = 120, cols = 160;
  std::vector<float> depth((size_t)rows * cols, 2.0f);
  float* d_depth; hipMalloc(&d_depth, depth.size() * 4); hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice);
  const nvbx_camera cam = {80.f, 80.f, 79.5f, 59.5f, cols, rows};
  const float T[16] = {0, 0, 1, 0,  -1, 0, 0, 0,  0, -1, 0, 0.4f,  0, 0, 0, 1};
  CHECK(nvbx_integrate_depth(m, d_depth, rows, cols, T, &cam));
  CHECK(nvbx_update_esdf(m));
  nvbx_device_view view;
  CHECK(nvbx_get_device_view(m, &view));
  const int3 mn = make_int3(0, -24, 0), sz = make_int3(56, 48, 16);
  const int64_t n = (int64_t)sz.x * sz.y * sz.z;
  float *d_esdf, *d_w; hipMalloc(&d_esdf, n * 4); hipMalloc(&d_w, n * 4);
  hipLaunchKernelGGL(k_sample, dim3(256), dim3(256), 0, stream, view, mn, sz, 1000.0f, d_esdf, d_w);
  // the library's own dense query of the same box (esdf_and_gradients_conversions.cu:88-125)
  float* d_ref; hipMalloc(&d_ref, n * 4);
  const int32_t mnv[3] = {mn.x, mn.y, mn.z}, szv[3] = {sz.x, sz.y, sz.z};
  CHECK(nvbx_esdf_dense_grid(m, mnv, szv, 1000.0f, d_ref));
  CHECK(nvbx_synchronize(m));
  std::vector<float> a(n), b(n), w(n);
  hipMemcpy(a.data(), d_esdf, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_ref, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(w.data(), d_w, n * 4, hipMemcpyDeviceToHost);
  int64_t mismatches = 0, known = 0, observed = 0; double wsum = 0.0;
  for (int64_t i = 0; i < n; i++) { if (a[i] != b[i]) mismatches++; if (a[i] < 999.f) known++; if (w[i] > 0.f) { observed++; wsum += w[i]; } }
  std::printf("{\"voxels\": %lld, \"mismatches\": %lld, \"esdf_known\": %lld, \"tsdf_observed\": %lld, \"tsdf_weight_sum\": %.6f, \"tsdf_blocks\": %lld}\n",
              (long long)n, (long long)mismatches, (long long)known, (long long)observed, wsum, (long long)nvbx_num_blocks(m, NVBX_LAYER_TSDF));
  nvbx_mapper_destroy(m);
  return mismatches == 0 && known > 0 && observed > 0 ? 0 : 2;
}
