// ground.hip -- MultiMapper::ground_plane_estimator() (nvblox_node.cpp:1456,1474; parameters mapper_initialization.cpp:133-153:
// experimental_use_ground_plane_estimation, ground_points_candidates_min/max_z_m, ransac_distance_threshold_m, num_ransac_iterations).
// [U] restated: (1) the ground CANDIDATES are the upward zero crossings of the TSDF -- vertically adjacent observed voxels with
// d(z) <= 0 < d(z + 1), the crossing interpolated linearly along z -- whose height lies in [min_z, max_z]; (2) the ground plane is the
// RANSAC plane of those points (sampled triples, inliers within the distance threshold, normal oriented upwards).  Not on the hot path
// (the node's debug visualisation; off in every shipped configuration): one streaming launch over the TSDF blocks, the sampling on the host.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "nvbx_mapper.h"

using namespace nvbx;

// one wavefront per TSDF block: lane = (x, y) column = 64 contiguous bytes; the voxel above the column's top comes from the block above
__global__ __launch_bounds__(64) void k_tsdf_zero_crossings(DMap m, float vs, float min_z, float max_z, float min_weight, float4* out, int32_t cap) {
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int lane = threadIdx.x, vx = lane >> 3, vy = lane & 7;
  for (int32_t slot = blockIdx.x; slot < hw; slot += gridDim.x) {
    if (!(m.slot_flags[slot] & F_TSDF)) continue;                         // uniform
    const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
    const float4* col = reinterpret_cast<const float4*>(&m.tsdf[(size_t)slot * 512 + 64 * vx + 8 * vy]);
    float d[9], w[9];
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 v = col[q]; d[2 * q] = v.x; w[2 * q] = v.y; d[2 * q + 1] = v.z; w[2 * q + 1] = v.w; }
    const uint32_t up = any_slot(m, bx, by, bz + 1);                       // (TSDF pool of a slot without that layer is all-zero: weight 0)
    const float2 top = slot_ok(up) ? m.tsdf[(size_t)up * 512 + 64 * vx + 8 * vy] : make_float2(0.0f, 0.0f);
    d[8] = top.x; w[8] = top.y;
#pragma unroll
    for (int z = 0; z < 8; z++) {
      bool hit = w[z] >= min_weight && w[z + 1] >= min_weight && d[z] <= 0.0f && d[z + 1] > 0.0f;
      float pz = 0.0f;
      if (hit) {
        const float z_lo = voxel_center(bz, z, vs * 8.0f, vs);
        pz = z_lo + vs * NVBX_DIV(-d[z], d[z + 1] - d[z]);
        hit = pz >= min_z && pz <= max_z;
      }
      const u64 mask = __ballot(hit);
      if (!mask) continue;
      int32_t base = 0;
      if (lane == 0) base = atomicAdd(&m.counters[C_TMP], (int32_t)__popcll(mask));
      base = __shfl(base, 0);
      if (hit) {
        const int32_t p = base + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (p < cap) out[p] = make_float4(voxel_center(bx, vx, vs * 8.0f, vs), voxel_center(by, vy, vs * 8.0f, vs), pz, 0.0f);
      }
    }
  }
}
__global__ void k_zero_tmp3(DMap m) { m.counters[C_TMP] = 0; }

extern "C" int64_t nvbx_tsdf_zero_crossings(nvbx_mapper* m, float min_z_m, float max_z_m, float* points_xyz_host, int64_t capacity) {
  if (!m || capacity < 0 || (capacity > 0 && !points_xyz_host) || !(max_z_m >= min_z_m)) { set_error("nvbx_tsdf_zero_crossings: invalid argument"); return NVBX_E_INVALID; }
  if (m->p.projective_layer_type == 1) return 0;                          // an occupancy mapper has no TSDF
  // The usual caller asks twice -- for the count, then with room for it (GroundPlaneEstimator::update, the Python mirror): the sorted result of
  // the first call is kept and serves the second, as long as no other entry point has come between (join_side drops it) and the band is the same.
  if (!(m->zc_valid && m->zc_min == min_z_m && m->zc_max == max_z_m)) {
    if (m->join_side()) return NVBX_E_DEVICE;
    m->zc_valid = false; m->zc_points.clear();
    int64_t cap = std::min<int64_t>(m->capacity * 64, (int64_t)1 << 24);       // one crossing per column and block in practice; a column can hold up to four
    for (int attempt = 0; attempt < 2; attempt++) {
      if (m->staging_bytes < cap * 16) {
        NVBX_HIP(hipStreamSynchronize(m->stream));
        if (m->staging) NVBX_HIP(hipFree(m->staging));
        m->staging = nullptr; m->staging_bytes = 0;
        NVBX_HIP(hipMalloc(&m->staging, (size_t)cap * 16));
        m->staging_bytes = cap * 16;
      }
      NVBX_LAUNCH(m, k_zero_tmp3, dim3(1), dim3(1), m->d);
      NVBX_LAUNCH(m, k_tsdf_zero_crossings, dim3((unsigned)std::min<int64_t>(m->capacity, 4096)), dim3(64), m->d, m->p.voxel_size, min_z_m, max_z_m,
                  m->p.esdf_min_weight > 0.0f ? m->p.esdf_min_weight : 1e-4f, (float4*)m->staging, (int32_t)cap);
      if (m->fetch_counters()) return NVBX_E_DEVICE;
      const int64_t found = m->h_counters[C_TMP];
      if (found <= cap) break;
      // more crossings than the buffer holds: never a silent, order-dependent truncation -- once more with room for all of them
      if (attempt == 1 || found > ((int64_t)1 << 27)) { set_error("nvbx_tsdf_zero_crossings: more zero crossings than the staging buffer can hold"); return NVBX_E_CAPACITY; }
      cap = found;
    }
    const int64_t n = m->h_counters[C_TMP];
    std::vector<float> tmp((size_t)n * 4);
    if (n) NVBX_HIP(hipMemcpy(tmp.data(), m->staging, (size_t)n * 16, hipMemcpyDeviceToHost));
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; i++) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {      // deterministic order: x, then y, then z
      const float* p = &tmp[(size_t)a * 4]; const float* q = &tmp[(size_t)b * 4];
      if (p[0] != q[0]) return p[0] < q[0]; if (p[1] != q[1]) return p[1] < q[1]; return p[2] < q[2]; });
    m->zc_points.resize((size_t)n * 3);
    for (int64_t i = 0; i < n; i++) { const float* p = &tmp[(size_t)order[(size_t)i] * 4]; m->zc_points[3 * i] = p[0]; m->zc_points[3 * i + 1] = p[1]; m->zc_points[3 * i + 2] = p[2]; }
    m->zc_valid = true; m->zc_min = min_z_m; m->zc_max = max_z_m;
  }
  const int64_t n = (int64_t)m->zc_points.size() / 3;
  if (n > capacity) return n;                                             // too small: the caller comes back with room for n
  if (n) memcpy(points_xyz_host, m->zc_points.data(), (size_t)n * 12);
  return n;
}

// [U] RansacPlaneFitter restated (host): `iterations` triples drawn with a fixed linear congruential sequence (Numerical Recipes' 1664525 /
// 1013904223; the estimate is a pure function of the points and the seed), plane through the triple with the normal turned upwards,
// inliers = points within `distance_threshold_m`; the plane with the most inliers (the earliest on ties) wins.  plane_out = {nx, ny, nz, d}
// with n . p + d = 0.  Returns the winner's inlier count; 0 = no plane (fewer than three points, or only degenerate triples).
extern "C" int64_t nvbx_fit_plane_ransac(const float* points_xyz, int64_t n, float distance_threshold_m, int32_t iterations, uint32_t seed, float plane_out[4]) {
  if (!points_xyz || !plane_out || n < 0 || iterations < 0 || !(distance_threshold_m >= 0.0f)) { set_error("nvbx_fit_plane_ransac: invalid argument"); return NVBX_E_INVALID; }
  plane_out[0] = 0.0f; plane_out[1] = 0.0f; plane_out[2] = 1.0f; plane_out[3] = 0.0f;
  if (n < 3) return 0;
  uint32_t state = seed;
  auto draw = [&]() { state = state * 1664525u + 1013904223u; return (int64_t)((state >> 8) % (uint32_t)std::min<int64_t>(n, 1 << 24)); };
  int64_t best = 0;
  for (int32_t it = 0; it < iterations; it++) {
    const int64_t i0 = draw(), i1 = draw(), i2 = draw();
    if (i0 == i1 || i0 == i2 || i1 == i2) continue;
    const float* a = points_xyz + 3 * i0; const float* b = points_xyz + 3 * i1; const float* c = points_xyz + 3 * i2;
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    if (!(len > 1e-12f)) continue;
    nx = nx / len; ny = ny / len; nz = nz / len;
    if (nz < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    const float d = -((nx * a[0] + ny * a[1]) + nz * a[2]);
    int64_t inl = 0;
    for (int64_t k = 0; k < n; k++) {
      const float* p = points_xyz + 3 * k;
      if (fabsf(((nx * p[0] + ny * p[1]) + nz * p[2]) + d) <= distance_threshold_m) inl++;
    }
    if (inl > best) { best = inl; plane_out[0] = nx; plane_out[1] = ny; plane_out[2] = nz; plane_out[3] = d; }
  }
  return best;
}
