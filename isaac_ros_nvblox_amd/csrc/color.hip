// color.hip -- MultiMapper::integrateColor on MI355X.
//
// Two launches per colour frame:
//   k_sphere_trace    one wavefront per 8x8 tile of the 1/f-resolution synthetic depth image; each lane sphere-traces
//                     its ray through the TSDF, caching the last block's slot so consecutive samples of a ray skip the
//                     hash probe and the last voxel's value so repeated samples of one voxel skip memory ([U] SphereTracer::cast restated).
//   k_integrate_color one 512-thread workgroup per allocated block slot (grid-stride over the slot range): frustum test
//                     (8 lanes = 8 corners), truncation-band test (block-wide vote), then the per-voxel projective
//                     colour blend -- selection and integration fused, no block list round trip
//                     ([U] ProjectiveColorIntegrator::integrateFrame restated).
// Call site served: nvblox_ros/src/lib/nvblox_node.cpp:1264.
#include <algorithm>
#include "nvbx_mapper.h"

using namespace nvbx;

// Slot of the block at (x,y,z) for a TSDF read, or SLOT_NONE.  No layer-flag check: the TSDF pool of a slot that does
// not carry F_TSDF is all-zero (freed / ESDF-only slots are zeroed, maintenance.hip), and weight 0 reads as "unobserved"
// exactly like a missing block.  One 16-B entry load per probe.
__device__ inline uint32_t tsdf_slot_any(const DMap& m, int32_t x, int32_t y, int32_t z) {
  const u64 key = pack_key(x, y, z);
  uint32_t h = table_pos(m, x, y, z);
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&m.table[h]);
    const u64 k = ((u64)e.y << 32) | (u64)e.x;
    if (k == key) return slot_ok(e.z) ? e.z : SLOT_NONE;
    if (k == KEY_EMPTY) return SLOT_NONE;
    h = (h + 1) & m.mask;
  }
  return SLOT_NONE;
}

// [U] SphereTracer::cast restated.  The march t <- t + tsdf(t) reads the TSDF by nearest voxel, so close to the surface
// it takes several steps inside ONE voxel (step = that voxel's small distance): a one-voxel register cache serves those
// without touching memory, and a one-block cache skips the hash probe while the ray stays inside a block.  The sequence
// of t values is bit-identical to the uncached march (same float operations in the same order).
__global__ __launch_bounds__(64) void k_sphere_trace(DMap m, Frame f, float* synth, int32_t srows, int32_t scols, int32_t max_steps,
                                                     float max_len, float eps_m) {
  const int lane = threadIdx.x;
  if (blockIdx.x == 0 && lane == 0) m.counters[C_COLOR_COUNT] = 0;
  const int tiles_x = (scols + 7) >> 3;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int r = ty * 8 + (lane >> 3), c = tx * 8 + (lane & 7);
  if (r >= srows || c >= scols) return;
  const float rx = (((float)(c * f.subsample) + 0.5f) - f.cu) / f.fu;
  const float ry = (((float)(r * f.subsample) + 0.5f) - f.cv) / f.fv;
  const float n = sqrtf((rx * rx + ry * ry) + 1.0f);
  const float dcx = rx / n, dcy = ry / n, dcz = 1.0f / n;
  float dl[3];
  rotate(f.R_LC, dcx, dcy, dcz, dl);
  bool last_positive = false, hit = false;
  float t = 0.0f;
  int32_t cbx = INT32_MIN, cby = 0, cbz = 0; uint32_t cslot = SLOT_NONE;      // block cache
  int32_t cgx = INT32_MIN, cgy = 0, cgz = 0; float2 cv = make_float2(0.0f, 0.0f);   // voxel cache
  for (int i = 0; i < max_steps && t < max_len; i++) {
    const float px = f.t_LC[0] + t * dl[0], py = f.t_LC[1] + t * dl[1], pz = f.t_LC[2] + t * dl[2];
    const int32_t gx = (int32_t)floorf(px / f.voxel_size), gy = (int32_t)floorf(py / f.voxel_size), gz = (int32_t)floorf(pz / f.voxel_size);
    if (gx != cgx || gy != cgy || gz != cgz) {
      const int32_t bx = gx >> 3, by = gy >> 3, bz = gz >> 3;
      if (bx != cbx || by != cby || bz != cbz) { cslot = tsdf_slot_any(m, bx, by, bz); cbx = bx; cby = by; cbz = bz; }
      cv = slot_ok(cslot) ? m.tsdf[(size_t)cslot * 512 + (gz & 7) + 8 * (gy & 7) + 64 * (gx & 7)] : make_float2(0.0f, 0.0f);
      cgx = gx; cgy = gy; cgz = gz;
    }
    float step;
    if (!(cv.y > 1e-4f)) {                       // missing block or unobserved voxel
      if (!last_positive) step = f.trunc; else break;
    } else {
      if (cv.x < eps_m) {
        if (last_positive) { t = t + cv.x; hit = true; }
        break;
      }
      step = cv.x; last_positive = true;
    }
    t = t + step;
  }
  synth[(int64_t)r * scols + c] = hit ? t * dcz : 0.0f;
}

__device__ inline uint32_t blend_u8(float c0, float w0, float c1, float w1) {
  const float tw = w0 + w1;
  const float a = w0 / tw, b = w1 / tw;
  float v = c0 * a + c1 * b;
  v = floorf(v + 0.5f);
  if (v < 0.0f) v = 0.0f;
  if (v > 255.0f) v = 255.0f;
  return (uint32_t)v;
}

struct SynthImg { const float* p; __device__ float operator()(int64_t i) const { return p[i]; } };

__global__ __launch_bounds__(512) void k_integrate_color(DMap m, Frame f, const uint8_t* rgb, const float* synth, int32_t srows, int32_t scols,
                                                         int32_t* color_list, int32_t* mesh_dirty, int32_t mesh_cnt) {
  __shared__ int s_out[6];
  __shared__ int s_band;
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  for (int32_t slot = blockIdx.x; slot < hw; slot += gridDim.x) {
    if (!(m.slot_flags[slot] & F_TSDF)) continue;     // uniform
    const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
    __syncthreads();
    if (tid < 6) s_out[tid] = 0;
    if (tid == 6) s_band = 0;
    __syncthreads();
    if (tid < 8) {   // frustum: count corners outside each plane
      float pc[3];
      apply_rt(f.R_CL, f.t_CL, (float)(bx + (tid & 1)) * f.block_size, (float)(by + ((tid >> 1) & 1)) * f.block_size,
               (float)(bz + ((tid >> 2) & 1)) * f.block_size, pc);
      if (f.fu * pc[0] + f.cu * pc[2] < 0.0f) atomicAdd(&s_out[0], 1);
      if (f.fu * pc[0] + (f.cu - (float)f.w) * pc[2] > 0.0f) atomicAdd(&s_out[1], 1);
      if (f.fv * pc[1] + f.cv * pc[2] < 0.0f) atomicAdd(&s_out[2], 1);
      if (f.fv * pc[1] + (f.cv - (float)f.h) * pc[2] > 0.0f) atomicAdd(&s_out[3], 1);
      if (pc[2] < 0.0f) atomicAdd(&s_out[4], 1);
      if (f.max_dist > 0.0f && pc[2] > f.max_dist) atomicAdd(&s_out[5], 1);
    }
    const float2 tv = m.tsdf[(size_t)slot * 512 + tid];
    if (tv.y > 1e-4f && fabsf(tv.x) < f.trunc) s_band = 1;   // benign race: all writers store 1
    __syncthreads();
    bool in_view = true;
#pragma unroll
    for (int q = 0; q < 6; q++) if (s_out[q] == 8) in_view = false;
    if (!in_view || !s_band) continue;                // uniform
    if (tid == 0) {
      const uint32_t old = atomicOr(&m.slot_flags[slot], F_COLOR | F_DIRTY_MESH);
      if (!(old & F_DIRTY_MESH)) mesh_dirty[atomicAdd(&m.counters[mesh_cnt], 1)] = slot;
      color_list[atomicAdd(&m.counters[C_COLOR_COUNT], 1)] = slot;
    }
    float pc[3];
    apply_rt(f.R_CL, f.t_CL, voxel_center(bx, vx, f.block_size, f.voxel_size), voxel_center(by, vy, f.block_size, f.voxel_size),
             voxel_center(bz, vz, f.block_size, f.voxel_size), pc);
    float u, v;
    if (!cam_project(f, pc, &u, &v)) continue;
    const float vd = pc[2];
    if (f.max_dist > 0.0f && vd > f.max_dist) continue;
    float sd;
    if (!interp_depth(SynthImg{synth}, srows, scols, u / (float)f.subsample, v / (float)f.subsample, 0, &sd)) continue;
    if (fabsf(sd - vd) > f.trunc) continue;
    // bilinear colour (interpolate2DLinear<Color>)
    const float uc = u - 0.5f, vc = v - 0.5f;
    const float fx = floorf(uc), fy = floorf(vc);
    const int x0 = (int)fx, y0 = (int)fy;
    if (x0 < 0 || y0 < 0 || x0 + 1 > f.cols - 1 || y0 + 1 > f.rows - 1) continue;
    const float ax = uc - fx, ay = vc - fy;
    const uint8_t* p00 = rgb + ((int64_t)y0 * f.cols + x0) * 3;
    const uint8_t* p01 = p00 + (int64_t)f.cols * 3;
    float c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float f00 = (float)p00[ch], f10 = (float)p00[3 + ch], f01 = (float)p01[ch], f11 = (float)p01[3 + ch];
      const float top = (1.0f - ax) * f00 + ax * f10;
      const float bot = (1.0f - ax) * f01 + ax * f11;
      c[ch] = (1.0f - ay) * top + ay * bot;
    }
    uint2* cp = &m.color[(size_t)slot * 512 + tid];
    const uint2 cur = *cp;
    const float w0 = __uint_as_float(cur.y);
    const uint32_t r8 = blend_u8((float)(cur.x & 0xFF), w0, c[0], 1.0f);
    const uint32_t g8 = blend_u8((float)((cur.x >> 8) & 0xFF), w0, c[1], 1.0f);
    const uint32_t b8 = blend_u8((float)((cur.x >> 16) & 0xFF), w0, c[2], 1.0f);
    *cp = make_uint2(r8 | (g8 << 8) | (b8 << 16), __float_as_uint(fminf(w0 + 1.0f, f.max_weight)));
  }
}

extern "C" int nvbx_integrate_color(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                    const nvbx_camera* camera) {
  if (!m || !rgb_dev || !T_L_C || !camera || rows <= 0 || cols <= 0) { set_error("nvbx_integrate_color: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  Frame f = m->make_frame(T_L_C, camera, rows, cols, m->p.sphere_tracing_subsampling);
  const int32_t srows = rows / f.subsample, scols = cols / f.subsample;
  if (srows < 2 || scols < 2) { set_error("colour image too small for the sphere-tracing subsampling"); return NVBX_E_INVALID; }
  if ((int64_t)srows * scols > m->synth_cap) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->synth) NVBX_HIP(hipFree(m->synth));
    m->synth = nullptr; m->synth_cap = 0;
    NVBX_HIP(hipMalloc(&m->synth, (size_t)srows * scols * 4));
    m->synth_cap = (int64_t)srows * scols;
  }
  m->synth_rows = srows; m->synth_cols = scols;
  const int tiles = ((srows + 7) / 8) * ((scols + 7) / 8);
  NVBX_LAUNCH(m, k_sphere_trace, dim3(tiles), dim3(64), m->d, f, m->synth, srows, scols, m->p.sphere_tracing_max_steps,
                     m->p.sphere_tracing_max_ray_length_m, m->p.sphere_tracing_surface_eps_vox * m->p.voxel_size);
  const int grid = (int)std::min<int64_t>(m->capacity, 2048);
  NVBX_LAUNCH(m, k_integrate_color, dim3(grid), dim3(512), m->d, f, rgb_dev, m->synth, srows, scols, m->color_list, m->mesh_dirty_live(), m->mesh_dirty_counter());
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

extern "C" int nvbx_get_synthetic_depth(nvbx_mapper* m, float* out_host, int64_t capacity, int32_t* rows, int32_t* cols) {
  if (!m || !rows || !cols) return NVBX_E_INVALID;
  *rows = m->synth_rows; *cols = m->synth_cols;
  const int64_t n = (int64_t)m->synth_rows * m->synth_cols;
  if (!out_host || n == 0) return NVBX_OK;
  if (n > capacity) return NVBX_E_CAPACITY;
  NVBX_HIP(hipMemcpyAsync(out_host, m->synth, n * 4, hipMemcpyDeviceToHost, m->stream));
  NVBX_HIP(hipStreamSynchronize(m->stream));
  return NVBX_OK;
}
