// convert.hip -- the small conversions that sit right beside the hot path in the reference node: depth image -> point
// cloud (DepthImageBackProjector), point-cloud transform, the two-layer costmap slice, and the device-side view handed
// to the caller's own kernels.  Call sites: nvblox_ros/src/lib/nvblox_node.cpp:836-840,1128-1131; fuser_node.cpp:294-297.
#include <algorithm>
#include <cmath>
#include "nvbx_mapper.h"
#include "nvbx_motion_math.h"
#include "nvbx_mask_geom.h"

using namespace nvbx;

// One pixel per thread, wave-aggregated compaction (one returning atomic per wavefront).
__global__ __launch_bounds__(256) void k_backproject(const float* depth, int32_t rows, int32_t cols, float fu, float fv, float cu, float cv,
                                                      float max_d, float* out, int64_t cap, int32_t* count) {
  const int64_t n = (int64_t)rows * cols;
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = base + lane;
    float d = 0.0f;
    if (i < n) d = depth[i];
    const bool ok = i < n && d > 0.0f && !(max_d > 0.0f && d > max_d);
    const u64 mask = __ballot(ok);
    if (!mask) continue;
    int32_t b = 0;
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) b = atomicAdd(count, (int32_t)__popcll(mask));
    b = __shfl(b, leader);
    if (ok) {
      const int64_t pos = (int64_t)b + __popcll(mask & ((1ull << lane) - 1ull));
      if (pos < cap) {
        const int32_t r = (int32_t)(i / cols), c = (int32_t)(i - (int64_t)r * cols);
        const float rx = (((float)c + 0.5f) - cu) / fu, ry = (((float)r + 0.5f) - cv) / fv;
        out[3 * pos] = d * rx; out[3 * pos + 1] = d * ry; out[3 * pos + 2] = d;
      }
    }
  }
}

extern "C" int nvbx_backproject_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const nvbx_camera* camera,
                                      float max_distance_m, float* points_xyz_dev, int64_t capacity_points, int64_t* n_points) {
  if (!m || !depth_dev || !camera || !points_xyz_dev || !n_points || rows <= 0 || cols <= 0 || capacity_points < 0) {
    set_error("nvbx_backproject_depth: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_HIP(hipMemsetAsync(m->export_count, 0, 4, m->stream));
  const int64_t n = (int64_t)rows * cols;
  NVBX_LAUNCH(m, k_backproject, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), depth_dev, rows, cols, camera->fu, camera->fv,
              camera->cu, camera->cv, max_distance_m, points_xyz_dev, capacity_points, m->export_count);
  int32_t cnt = 0;
  NVBX_HIP(hipMemcpyAsync(&cnt, m->export_count, 4, hipMemcpyDeviceToHost, m->stream));
  NVBX_HIP(hipStreamSynchronize(m->stream));
  *n_points = std::min<int64_t>(cnt, capacity_points);
  if (cnt > capacity_points) { set_error("nvbx_backproject_depth: point capacity too small"); return NVBX_E_CAPACITY; }
  return NVBX_OK;
}

__global__ __launch_bounds__(256) void k_transform_points(Rt T, const float* in, int64_t n, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float o[3];
    apply_rt(T.R, T.t, in[3 * i], in[3 * i + 1], in[3 * i + 2], o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
  }
}
extern "C" int nvbx_transform_pointcloud(nvbx_mapper* m, const float T_L_C[16], const float* points_in_dev, int64_t n_points, float* points_out_dev) {
  if (!m || !T_L_C || n_points < 0 || (n_points > 0 && (!points_in_dev || !points_out_dev))) { set_error("nvbx_transform_pointcloud: invalid argument"); return NVBX_E_INVALID; }
  if (n_points == 0) return NVBX_OK;
  NVBX_HIP(hipSetDevice(m->device));
  Rt T;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[3 * i + j] = T_L_C[4 * i + j]; T.t[i] = T_L_C[4 * i + 3]; }
  NVBX_LAUNCH(m, k_transform_points, dim3((unsigned)std::min<int64_t>((n_points + 255) / 256, 2048)), dim3(256), T, points_in_dev, n_points, points_out_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ LiDAR motion compensation
__global__ __launch_bounds__(256) void k_motion_compensate(nvbx_rel_motion mo, const float* in, const float* rel_ms, int64_t n, float inv_duration, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float a = rel_ms[i] * inv_duration;
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    const float p[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
    float o[3]; nvbx_motion_compensate_point(&mo, a, p, o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
  }
}
extern "C" int nvbx_motion_compensate_pointcloud(nvbx_mapper* m, const float* points_in_dev, const float* rel_time_ms_dev, int64_t n_points,
                                                 const float T_L_S_start[16], const float T_L_S_end[16], float scan_duration_ms, float* points_out_dev) {
  if (!m || !T_L_S_start || !T_L_S_end || n_points < 0 || (n_points > 0 && (!points_in_dev || !rel_time_ms_dev || !points_out_dev)) || !(scan_duration_ms > 0.0f)) {
    set_error("nvbx_motion_compensate_pointcloud: invalid argument"); return NVBX_E_INVALID; }
  if (n_points == 0) return NVBX_OK;
  NVBX_HIP(hipSetDevice(m->device));
  const nvbx_rel_motion mo = nvbx_rel_motion_make(T_L_S_start, T_L_S_end);
  NVBX_LAUNCH(m, k_motion_compensate, dim3((unsigned)std::min<int64_t>((n_points + 255) / 256, 2048)), dim3(256), mo, points_in_dev, rel_time_ms_dev, n_points,
              1.0f / scan_duration_ms, points_out_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ two-layer slice
__device__ inline float slice_value(const DMap& m, int32_t bx, int32_t by, int32_t bz_out, int32_t vz_out, int lane, float voxel_size, bool* known) {
  *known = false;
  const uint32_t es = find_slot(m, bx, by, bz_out, F_ESDF);
  if (!slot_ok(es)) return 0.0f;
  const uint2 e = m.esdf[(size_t)es * 512 + vz_out * 64 + lane];
  if (!(e.y & ESDF_OBSERVED)) return 0.0f;
  *known = true;
  const float v = sqrtf(__uint_as_float(e.x)) * voxel_size;
  return (e.y & ESDF_INSIDE) ? -v : v;
}
// one wavefront per 8x8 block of the image (lane = pixel of the block's slice plane)
__global__ __launch_bounds__(64) void k_esdf_slice_combined(DMap m1, DMap m2, int32_t bz1, int32_t vz1, int32_t bz2, int32_t vz2, float vs1, float vs2,
                                                            float unknown, float* img, int32_t bx0, int32_t by0, int32_t nbx, int32_t nby) {
  const int lane = threadIdx.x;
  const int vx = lane & 7, vy = lane >> 3;
  const int32_t cols = nbx * 8;
  for (int32_t c = blockIdx.x; c < nbx * nby; c += gridDim.x) {
    const int32_t cy = c / nbx, cx = c - cy * nbx;
    bool k1, k2;
    const float d1 = slice_value(m1, bx0 + cx, by0 + cy, bz1, vz1, lane, vs1, &k1);
    const float d2 = slice_value(m2, bx0 + cx, by0 + cy, bz2, vz2, lane, vs2, &k2);
    float v = unknown;
    if (k1 && k2) v = fminf(d1, d2); else if (k1) v = d1; else if (k2) v = d2;
    img[(int64_t)(cy * 8 + vy) * cols + cx * 8 + vx] = v;
  }
}

static int combined_extent(nvbx_mapper* m1, nvbx_mapper* m2, int32_t c[4]) {
  if (m1->device != m2->device) { set_error("combined slice: the two mappers live on different devices"); return NVBX_E_INVALID; }
  if (fabsf(m1->p.voxel_size - m2->p.voxel_size) > 1e-6f * m1->p.voxel_size) { set_error("combined slice: the two mappers have different voxel sizes"); return NVBX_E_INVALID; }
  if (m1->fetch_counters() || m2->fetch_counters()) return NVBX_E_DEVICE;      // (synchronises both streams)
  const int32_t* a = m1->h_counters + C_ESDF_AABB; const int32_t* b = m2->h_counters + C_ESDF_AABB;
  c[0] = std::min(a[0], b[0]); c[1] = std::min(a[1], b[1]); c[2] = std::max(a[2], b[2]); c[3] = std::max(a[3], b[3]);
  return NVBX_OK;
}
extern "C" int nvbx_esdf_slice_combined_size(nvbx_mapper* m1, nvbx_mapper* m2, int32_t* rows, int32_t* cols, float aabb[6]) {
  if (!m1 || !m2 || !rows || !cols) return NVBX_E_INVALID;
  int32_t c[4];
  const int rc = combined_extent(m1, m2, c); if (rc) return rc;
  if (c[0] > c[2]) { *rows = 0; *cols = 0; return NVBX_OK; }
  const EsdfArgs a = m1->make_esdf_args();
  const float bs = m1->p.voxel_size * 8.0f;
  *cols = (c[2] - c[0] + 1) * 8; *rows = (c[3] - c[1] + 1) * 8;
  if (aabb) {
    aabb[0] = (float)c[0] * bs; aabb[1] = (float)c[1] * bs; aabb[2] = (float)a.bz_out * bs;
    aabb[3] = (float)(c[2] + 1) * bs; aabb[4] = (float)(c[3] + 1) * bs; aabb[5] = (float)(a.bz_out + 1) * bs;
  }
  return NVBX_OK;
}
extern "C" int nvbx_esdf_slice_combined_to_image(nvbx_mapper* m1, nvbx_mapper* m2, float unknown_value, float* image_dev, int64_t capacity_elems,
                                                 int32_t* rows, int32_t* cols, float aabb[6]) {
  if (!m1 || !m2 || !rows || !cols) return NVBX_E_INVALID;
  int rc = nvbx_esdf_slice_combined_size(m1, m2, rows, cols, aabb); if (rc) return rc;     // both streams are idle after this
  if (*rows == 0) return NVBX_OK;
  if (!image_dev || (int64_t)*rows * *cols > capacity_elems) { set_error("slice image capacity too small"); return NVBX_E_CAPACITY; }
  int32_t c[4];
  { const int32_t* a = m1->h_counters + C_ESDF_AABB; const int32_t* b = m2->h_counters + C_ESDF_AABB;
    c[0] = std::min(a[0], b[0]); c[1] = std::min(a[1], b[1]); c[2] = std::max(a[2], b[2]); c[3] = std::max(a[3], b[3]); }
  const EsdfArgs a1 = m1->make_esdf_args(), a2 = m2->make_esdf_args();
  const int32_t nbx = c[2] - c[0] + 1, nby = c[3] - c[1] + 1;
  NVBX_HIP(hipSetDevice(m1->device));
  NVBX_LAUNCH(m1, k_esdf_slice_combined, dim3(std::min(nbx * nby, 4096)), dim3(64), m1->d, m2->d, a1.bz_out, a1.vz_out, a2.bz_out, a2.vz_out,
              m1->p.voxel_size, m2->p.voxel_size, unknown_value, image_dev, c[0], c[1], nbx, nby);
  NVBX_HIP(hipGetLastError());
  // m2 must not change its map before the kernel has read it: it is a reader on m1's stream, so m2's stream waits for it
  if (m2->stream != m1->stream) {
    hipEvent_t e = m1->get_event();
    NVBX_HIP(hipEventRecord(e, m1->stream));
    NVBX_HIP(hipStreamWaitEvent(m2->stream, e, 0));
    m1->event_pool.push_back(e);             // (the wait captured the recorded state; the event can be reused)
  }
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ mask splitting
__global__ __launch_bounds__(256) void k_mask_zmin(MaskGeom g, const float* depth, uint32_t* zmin) {
  const int64_t n = (int64_t)g.rows * g.cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = depth[i];
    if (!(d > 0.0f)) continue;
    float z; const int32_t mi = mask_pixel(g, (int32_t)(i / g.cols), (int32_t)(i % g.cols), d, &z);
    if (mi >= 0) atomicMin(&zmin[mi], __float_as_uint(z));       // positive floats order like their bit patterns
  }
}
__global__ __launch_bounds__(256) void k_split_depth(MaskGeom g, const float* depth, const uint8_t* mask, const uint32_t* zmin, float thr,
                                                     float* unmasked, float* masked, uint8_t* overlay) {
  const int64_t n = (int64_t)g.rows * g.cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = depth[i];
    bool is_masked = false;
    if (d > 0.0f) {
      float z; const int32_t mi = mask_pixel(g, (int32_t)(i / g.cols), (int32_t)(i % g.cols), d, &z);
      if (mi >= 0 && mask[mi] != 0 && z <= __uint_as_float(zmin[mi]) + thr) is_masked = true;
    }
    unmasked[i] = (d > 0.0f && is_masked) ? NVBX_MASKED_DEPTH_INVALID : d;
    masked[i] = (d > 0.0f && is_masked) ? d : NVBX_MASKED_DEPTH_INVALID;
    if (overlay) {
      float gv = d > 0.0f ? d * (255.0f / 5.0f) : 0.0f; if (gv > 255.0f) gv = 255.0f;
      const uint8_t q = (uint8_t)gv;
      overlay[3 * i] = is_masked ? 255 : q; overlay[3 * i + 1] = q; overlay[3 * i + 2] = q;
    }
  }
}
extern "C" int nvbx_split_depth_by_mask(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const uint8_t* mask_dev,
                                        int32_t mask_rows, int32_t mask_cols, const float T_CM_CD[16], const nvbx_camera* dc,
                                        const nvbx_camera* mc, float occlusion_threshold_m, float* unmasked_dev, float* masked_dev, uint8_t* overlay_dev) {
  if (!m || !depth_dev || !mask_dev || !T_CM_CD || !dc || !mc || !unmasked_dev || !masked_dev || rows <= 0 || cols <= 0 || mask_rows <= 0 || mask_cols <= 0) {
    set_error("nvbx_split_depth_by_mask: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  const int64_t mn = (int64_t)mask_rows * mask_cols;
  if (mn > m->mask_zmin_cap) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->mask_zmin) NVBX_HIP(hipFree(m->mask_zmin));
    m->mask_zmin = nullptr; m->mask_zmin_cap = 0;
    NVBX_HIP(hipMalloc(&m->mask_zmin, (size_t)mn * 4));
    m->mask_zmin_cap = mn;
  }
  const MaskGeom g = make_mask_geom(T_CM_CD, dc, mc, rows, cols, mask_rows, mask_cols);
  NVBX_HIP(hipMemsetAsync(m->mask_zmin, 0x7F, (size_t)mn * 4, m->stream));        // 0x7F7F7F7F = 3.4e38
  const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)rows * cols + 255) / 256, 2048);
  NVBX_LAUNCH(m, k_mask_zmin, dim3(grid), dim3(256), g, depth_dev, m->mask_zmin);
  NVBX_LAUNCH(m, k_split_depth, dim3(grid), dim3(256), g, depth_dev, mask_dev, (const uint32_t*)m->mask_zmin, occlusion_threshold_m, unmasked_dev, masked_dev, overlay_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
__global__ __launch_bounds__(256) void k_split_color(const uint8_t* rgb, const uint8_t* mask, int64_t n, uint8_t* unmasked, uint8_t* masked) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool mk = mask[i] != 0;
    const uint8_t r = rgb[3 * i], gg = rgb[3 * i + 1], b = rgb[3 * i + 2];
    if (unmasked) { unmasked[3 * i] = mk ? 0 : r; unmasked[3 * i + 1] = mk ? 0 : gg; unmasked[3 * i + 2] = mk ? 0 : b; }
    if (masked) { masked[3 * i] = mk ? r : 0; masked[3 * i + 1] = mk ? gg : 0; masked[3 * i + 2] = mk ? b : 0; }
  }
}
extern "C" int nvbx_split_color_by_mask(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const uint8_t* mask_dev,
                                        uint8_t* rgb_unmasked_dev, uint8_t* rgb_masked_dev) {
  if (!m || !rgb_dev || !mask_dev || rows <= 0 || cols <= 0 || (!rgb_unmasked_dev && !rgb_masked_dev)) { set_error("nvbx_split_color_by_mask: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  const int64_t n = (int64_t)rows * cols;
  NVBX_LAUNCH(m, k_split_color, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), rgb_dev, mask_dev, n, rgb_unmasked_dev, rgb_masked_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ device view
extern "C" int nvbx_get_device_view(nvbx_mapper* m, nvbx_device_view* out) {
  if (!m || !out) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;           // a kernel launched with the view sees the completed ESDF update
  out->table = m->d.table; out->table_mask = m->d.mask; out->table_shift = m->d.shift;
  out->slot_flags = m->d.slot_flags; out->slot_index = m->d.slot_index;
  out->tsdf = m->d.tsdf; out->color = m->d.color; out->esdf = m->d.esdf;
  out->voxel_size = m->p.voxel_size; out->block_capacity = (uint32_t)m->capacity;
  return NVBX_OK;
}
