// mapper.hip -- lifetime, parameters, layer accessors of libnvblox_hip.so (host code + small utility kernels).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include "nvbx_mapper.h"

using namespace nvbx;

namespace nvbx {
static thread_local std::string g_err;
void set_error(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
}
void set_error(const char* what) { g_err = what; }
}  // namespace nvbx

extern "C" const char* nvbx_last_error(void) { return nvbx::g_err.c_str(); }

// ------------------------------------------------------------------------------------------------ utility kernels
__global__ void k_init_map(DMap m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m.capacity) { m.free_stack[i] = m.capacity - 1 - i; m.slot_flags[i] = 0; m.slot_stamp[i] = STAMP_NEVER; m.slot_consumed[i] = STAMP_NEVER; m.slot_cam[i] = STAMP_NEVER; }
  if (i < (uint32_t)(S_NUM * NSH * SH_STRIDE)) {       // sharded counters: zero, ESDF window records min = +inf / max = -inf
    const int id = (int)i / (NSH * SH_STRIDE), field = (int)i % SH_STRIDE;
    int32_t v = 0;
    if (id == S_ESDF_REC || id == S_ESDF_REC + 1) { if (field < 2) v = INT32_MAX; else if (field < 4) v = INT32_MIN; }
    m.shc[i] = v;
  }
  if (i < C_NUM) {
    int32_t v = 0;
    if (i == C_FREE_TOP) v = (int32_t)m.capacity;
    const int a = (int)i - C_ESDF_AABB;
    if (a >= 0 && a < 4) v = a < 2 ? INT32_MAX : INT32_MIN;
    const int w3 = (int)i - C_ESDF3_WIN;
    if (w3 >= 0 && w3 < 6) v = w3 < 3 ? INT32_MAX : INT32_MIN;
#ifdef NVBX_CHECK_INVARIANTS
    if (i >= C_INV_I1 && i <= C_INV_I3) v = m.counters[i];      // (violations survive clear(): a mapper answers for all its launches when it is closed)
#endif
    m.counters[i] = v;
  }
}

// collect Index3D of every live slot carrying `layer` (order arbitrary; host sorts)
__global__ void k_collect_indices(DMap m, uint32_t layer, int32_t* out, int32_t cap) {
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < hw; s += gridDim.x * blockDim.x) {
    if (m.slot_flags[s] & layer) {
      const int32_t p = atomicAdd(&m.counters[C_TMP], 1);
      if (p < cap) { out[3 * p] = m.slot_index[3 * s]; out[3 * p + 1] = m.slot_index[3 * s + 1]; out[3 * p + 2] = m.slot_index[3 * s + 2]; }
    }
  }
}
__global__ void k_zero_tmp(DMap m) { m.counters[C_TMP] = 0; }

// view list ({slot, x, y, z} records of one frame) -> Index3D.  cam_mask != 0 (the frame was a batch): only the blocks the
// cameras of the mask had in view -- "the last depth view" of a batch is its last camera's, as separate calls would leave it.
__global__ void k_viewlist_to_indices(DMap m, const int4* list, int32_t count_idx, uint32_t cam_mask, int32_t* out, int32_t cap) {
  int32_t n = m.counters[count_idx]; if (n > cap) n = cap;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int4 r = list[i];
    bool ok = slot_ok((uint32_t)r.x) && m.slot_flags[(uint32_t)r.x];
    if (ok && cam_mask && !(m.table[m.slot_entry[(uint32_t)r.x]].stamp & cam_mask)) ok = false;
    out[3 * i] = ok ? r.y : INT32_MIN; out[3 * i + 1] = ok ? r.z : INT32_MIN; out[3 * i + 2] = ok ? r.w : INT32_MIN;
  }
}

// sharded work list (slot ids) -> Index3D, written to out[0..n) in list order; *n_out = entries
__global__ void k_shardlist_to_indices(DMap m, int list, int32_t* out, int32_t cap, int32_t* n_out) {
  ListView v; int32_t n = list_open(m, list, &v); if (n > cap) n = cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = (uint32_t)list_at(m, list, v, i);
    if (slot_ok(s) && m.slot_flags[s]) { out[3 * i] = m.slot_index[3 * s]; out[3 * i + 1] = m.slot_index[3 * s + 1]; out[3 * i + 2] = m.slot_index[3 * s + 2]; }
    else { out[3 * i] = INT32_MIN; out[3 * i + 1] = INT32_MIN; out[3 * i + 2] = INT32_MIN; }
  }
}

// gather n blocks of `layer` into a dense buffer in the REFERENCE voxel struct layout (z + 8y + 64x order).
// found[i] = 1 if the block exists.  One 512-thread workgroup per block.
// `layer` is the INTERNAL flag; occupancy = 1: the projective pool holds log-odds and leaves as nvbx_occupancy_voxel {f32}
__global__ __launch_bounds__(512) void k_gather_blocks(DMap m, uint32_t layer, int32_t occupancy, const int32_t* idx, int32_t n, uint8_t* out, int32_t* found) {
  const int i = blockIdx.x; if (i >= n) return;
  const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], layer);
  const int t = threadIdx.x;
  if (t == 0) found[i] = slot_ok(s) ? 1 : 0;
  if (!slot_ok(s)) {            // absent block: zeros, never stale staging bytes
    const size_t vb = layer == F_ESDF ? sizeof(nvbx_esdf_voxel) : (layer == F_FREESPACE ? sizeof(nvbx_freespace_voxel) : ((layer == F_TSDF && occupancy) ? 4 : 8));
    for (size_t q = t; q < 512 * vb / 4; q += 512) reinterpret_cast<uint32_t*>(out + (size_t)i * 512 * vb)[q] = 0u;
    return;
  }
  if (layer == F_TSDF && occupancy) { reinterpret_cast<float*>(out)[(size_t)i * 512 + t] = m.tsdf[(size_t)s * 512 + t].x; }
  else if (layer == F_TSDF) { reinterpret_cast<float2*>(out)[(size_t)i * 512 + t] = m.tsdf[(size_t)s * 512 + t]; }
  else if (layer == F_COLOR) { reinterpret_cast<uint2*>(out)[(size_t)i * 512 + t] = m.color[(size_t)s * 512 + t]; }
  else if (layer == F_FREESPACE) {
    const int4 v = m.freespace[(size_t)s * 512 + t];
    nvbx_freespace_voxel o;
    o.last_occupied_timestamp_ms = (int64_t)(((u64)(uint32_t)v.y << 32) | (u64)(uint32_t)v.x);
    o.consecutive_occupancy_duration_ms = (int64_t)v.z;
    o.is_high_confidence_freespace = (uint8_t)(v.w & 1); o.initialized = (uint8_t)((v.w >> 1) & 1);
    for (int q = 0; q < 6; q++) o.pad[q] = 0;
    reinterpret_cast<nvbx_freespace_voxel*>(out)[(size_t)i * 512 + t] = o;
  }
  else if (layer == F_ESDF) {
    const int x = t >> 6, y = (t >> 3) & 7, z = t & 7;             // reference order
    const uint2 v = m.esdf[(size_t)s * 512 + x + 8 * y + 64 * z];  // device order
    nvbx_esdf_voxel o;
    o.squared_distance_vox = __uint_as_float(v.x);
    o.parent_direction[0] = (int8_t)(v.y & 0xFF); o.parent_direction[1] = (int8_t)((v.y >> 8) & 0xFF); o.parent_direction[2] = (int8_t)((v.y >> 16) & 0xFF);
    o.observed = (v.y & ESDF_OBSERVED) ? 1 : 0; o.is_inside = (v.y & ESDF_INSIDE) ? 1 : 0; o.is_site = (v.y & ESDF_SITE) ? 1 : 0; o.pad = 0;
    reinterpret_cast<nvbx_esdf_voxel*>(out)[(size_t)i * 512 + t] = o;
  }
}

// allocateBlockAtIndex + whole-block write from reference structs: workgroup i writes block idx[i] from in[i][512]
__global__ __launch_bounds__(512) void k_scatter_blocks(DMap m, uint32_t layer, int32_t occupancy, const int32_t* idx, const uint8_t* in_all, size_t block_bytes,
                                                        int32_t mesh_list, int32_t bz_out, int32_t vz_out, float trunc) {
  __shared__ uint32_t s_slot;
  __shared__ u64 s_sites, s_obs, s_ins;
  const int t = threadIdx.x;
  const int32_t x = idx[3 * blockIdx.x], y = idx[3 * blockIdx.x + 1], z = idx[3 * blockIdx.x + 2];
  const uint8_t* in = in_all + (size_t)blockIdx.x * block_bytes;
  if (t == 0) {
    bool is_new; const int32_t h = hash_insert(m, x, y, z, layer, &is_new);
    uint32_t s = SLOT_NONE;
    // (a duplicate index in one batch: the workgroup that lost the insert waits for the winner to publish the slot, as mark_block does)
    if (h >= 0) { do { s = ld_slot_acquire(&m.table[h]); } while (s == SLOT_INVALID); }
    if (slot_ok(s)) {
      uint32_t add = layer;
      if (layer == F_TSDF) add |= F_DIRTY_ESDF | F_DIRTY_MESH;
      const uint32_t old = atomicOr(&m.slot_flags[s], add);
      if (layer == F_TSDF) {
        if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)s);
        if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, (int32_t)s);
      }
      if (layer == F_ESDF && z == bz_out) {     // (3-D ESDF: only the slice plane's blocks span the slicer's image)
        atomicMin(&m.counters[C_ESDF_AABB + 0], x); atomicMin(&m.counters[C_ESDF_AABB + 1], y);
        atomicMax(&m.counters[C_ESDF_AABB + 2], x); atomicMax(&m.counters[C_ESDF_AABB + 3], y);
      }
    }
    s_slot = s; s_sites = 0ull; s_obs = 0ull; s_ins = 0ull;
  }
  __syncthreads();
  const uint32_t s = s_slot;
  if (!slot_ok(s)) return;
  if (layer == F_TSDF && occupancy) m.tsdf[(size_t)s * 512 + t] = make_float2(reinterpret_cast<const float*>(in)[t], 0.0f);
  else if (layer == F_TSDF) {
    const float2 v = reinterpret_cast<const float2*>(in)[t];
    m.tsdf[(size_t)s * 512 + t] = v;
    publish_band(m.slot_flags, s, t, in_band(v.x, v.y, trunc));        // (uniform branch: every wavefront is here)
  }
  else if (layer == F_COLOR) m.color[(size_t)s * 512 + t] = reinterpret_cast<const uint2*>(in)[t];
  else if (layer == F_ESDF) {
    const int vx = t >> 6, vy = (t >> 3) & 7, vz = t & 7;
    const nvbx_esdf_voxel v = reinterpret_cast<const nvbx_esdf_voxel*>(in)[t];
    m.esdf[(size_t)s * 512 + vx + 8 * vy + 64 * vz] =
        make_uint2(__float_as_uint(v.squared_distance_vox),
                   esdf_meta(v.parent_direction[0], v.parent_direction[1], v.parent_direction[2], v.observed, v.is_inside, v.is_site));
    // keep the slice plane's site mask (what k_esdf_edt reads) consistent with the written voxels
    if (z == bz_out && vz == vz_out) {
      if (v.is_site) atomicOr(&s_sites, 1ull << (vx + 8 * vy));
      if (v.observed) atomicOr(&s_obs, 1ull << (vx + 8 * vy));
      if (v.is_inside) atomicOr(&s_ins, 1ull << (vx + 8 * vy));
    }
    __syncthreads();
    if (t == 0) { m.site_bits[s] = (z == bz_out) ? s_sites : 0ull; m.obs_bits[s] = (z == bz_out) ? s_obs : 0ull; m.inside_bits[s] = (z == bz_out) ? s_ins : 0ull; }
  }
}

__global__ __launch_bounds__(512) void k_recompute_band(DMap m, float trunc) {
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t slot = blockIdx.x; slot < hw; slot += gridDim.x) {
    if (!(m.slot_flags[slot] & F_TSDF)) continue;             // uniform
    const float2 v = m.tsdf[(size_t)slot * 512 + threadIdx.x];
    publish_band(m.slot_flags, (uint32_t)slot, (int)threadIdx.x, in_band(v.x, v.y, trunc));
  }
}

// ------------------------------------------------------------------------------------------------ host helpers
static int alloc_all(nvbx_mapper* m) {
  const int64_t cap = m->capacity;
  uint64_t tsz = 1; while (tsz < (uint64_t)cap * 2) tsz <<= 1;
  DMap& d = m->d;
  d.capacity = (uint32_t)cap; d.mask = (uint32_t)(tsz - 1);
  { uint32_t lg = 0; while ((1ull << lg) < tsz) lg++; d.shift = 32u - lg; }
  NVBX_HIP(hipMalloc(&d.table, tsz * sizeof(Entry)));
  NVBX_HIP(hipMalloc(&d.free_stack, cap * 4));
  NVBX_HIP(hipMalloc(&d.counters, C_NUM * 4));
  // (k_init_map of the -DNVBX_CHECK_INVARIANTS variant keeps some of them across clear().  ON THE MAPPER'S STREAM: a memset on the null stream is
  //  not ordered with a non-blocking stream and landed after k_init_map once in a while -- a map with no free slot, tests/cpp rccl_fusion)
  NVBX_HIP(hipMemsetAsync(d.counters, 0, C_NUM * 4, m->stream));
  NVBX_HIP(hipMalloc(&d.slot_flags, cap * 4));
  NVBX_HIP(hipMalloc(&d.slot_index, cap * 12));
  NVBX_HIP(hipMalloc(&d.slot_entry, cap * 4));
  NVBX_HIP(hipMalloc(&d.slot_stamp, cap * 4));
  NVBX_HIP(hipMalloc(&d.slot_consumed, cap * 4));
  NVBX_HIP(hipMalloc(&d.slot_cam, cap * 4));
  NVBX_HIP(hipMalloc(&d.tsdf, cap * 4096));
  NVBX_HIP(hipMalloc(&d.color, cap * 4096));
  NVBX_HIP(hipMalloc(&d.esdf, cap * 4096));
  NVBX_HIP(hipMalloc(&m->view_list, cap * 16));    // int4 {slot, x, y, z} per block in view
  NVBX_HIP(hipMalloc(&d.lists, (size_t)N_LISTS * NSH * cap * 4));
  NVBX_HIP(hipMalloc(&d.shc, S_NUM * NSH * SH_STRIDE * 4));
  NVBX_HIP(hipMalloc(&m->export_idx, cap * 12));
  NVBX_HIP(hipMalloc(&m->export_count, 64));
  NVBX_HIP(hipMalloc(&m->cleared_idx, cap * 12));
  NVBX_HIP(hipMalloc(&d.site_bits, cap * 8));
  NVBX_HIP(hipMalloc(&d.obs_bits, cap * 8));
  NVBX_HIP(hipMalloc(&d.inside_bits, cap * 8));
  // mesh arena
  m->mesh_vert_cap = std::min<int64_t>(cap * 192, 48ll << 20); m->mesh_tri_cap = m->mesh_vert_cap * 2;
  NVBX_HIP(hipMalloc(&m->mesh_vert, m->mesh_vert_cap * 12));
  NVBX_HIP(hipMalloc(&m->mesh_nrm, m->mesh_vert_cap * 12));
  NVBX_HIP(hipMalloc(&m->mesh_col, m->mesh_vert_cap * 4));
  NVBX_HIP(hipMalloc(&m->mesh_tri, m->mesh_tri_cap * 12));
  NVBX_HIP(hipMalloc(&m->mesh_rec, cap * sizeof(MeshRecord)));
  m->staging_bytes = 8 << 20;
  NVBX_HIP(hipMalloc(&m->staging, m->staging_bytes));
  NVBX_HIP(hipHostMalloc(&m->h_counters, C_NUM * 4));
  NVBX_HIP(hipHostMalloc(&m->h_shc, S_NUM * NSH * SH_STRIDE * 4));
  NVBX_HIP(hipHostMalloc(&m->h_mirror, 64, hipHostMallocMapped));
  { void* dp = nullptr; NVBX_HIP(hipHostGetDevicePointer(&dp, m->h_mirror, 0)); d.host_mirror = (int32_t*)dp; }
  m->h_mirror[0] = (int32_t)cap; m->h_mirror[1] = 0; m->h_mirror[2] = 0; m->h_mirror[3] = 0; m->h_mirror[4] = 0; m->h_mirror[8] = 0;      // ([4]: fence progress of held-back colour frames, frames.hip -- never reset)
  return NVBX_OK;
}

static int reset_map(nvbx_mapper* m) {
  DMap& d = m->d;
  const int64_t cap = m->capacity;
  NVBX_HIP(hipMemsetAsync(d.table, 0xFF, ((size_t)d.mask + 1) * sizeof(Entry), m->stream));
  NVBX_HIP(hipMemsetAsync(d.tsdf, 0, cap * 4096, m->stream));
  NVBX_HIP(hipMemsetAsync(d.color, 0, cap * 4096, m->stream));
  NVBX_HIP(hipMemsetAsync(d.esdf, 0, cap * 4096, m->stream));
  NVBX_HIP(hipMemsetAsync(d.site_bits, 0, cap * 8, m->stream));
  NVBX_HIP(hipMemsetAsync(d.obs_bits, 0, cap * 8, m->stream));
  NVBX_HIP(hipMemsetAsync(d.inside_bits, 0, cap * 8, m->stream));
  NVBX_HIP(hipMemsetAsync(m->export_count, 0, 64, m->stream));
  if (d.freespace) NVBX_HIP(hipMemsetAsync(d.freespace, 0, cap * 512 * 16, m->stream));
  const int64_t n = std::max<int64_t>(cap, std::max<int64_t>(C_NUM, S_NUM * NSH * SH_STRIDE));
  NVBX_LAUNCH(m, k_init_map, dim3((unsigned)((n + 255) / 256)), dim3(256), d);
  NVBX_HIP(hipGetLastError());
  m->dirty_since_mark = false; m->premark_consumed = false; m->mark_pass = 0; m->edt_pending = false; m->import_pending = false;
  m->unresolved_marks = false; m->pass_at_last_edt = 0; (void)m->take_pending(); m->release_consumed_frames(); m->esdf_update_pending = false; m->lidar_integrated = false;
  if (m->h_mirror) { m->h_mirror[0] = (int32_t)m->capacity; m->h_mirror[1] = 0; m->h_mirror[2] = 0; m->h_mirror[3] = 0; }
  m->frame_id = 0; m->esdf_epoch = 0; m->mesh_epoch = 0; m->last_view_frame = 0; m->last_camera_view_frame = 0; m->synth_rows = m->synth_cols = 0;
  return NVBX_OK;
}

// (a polled stream write -- hipStreamWriteValue32 of a sequence number into pinned memory, the host spinning on it -- was measured in round 5 and not
//  kept: the write is itself ~15 us late, a frame waited for took 0.066 instead of 0.052 ms; EXPERIMENTS.md)
int nvbx_mapper::wait_stream() {
  NVBX_HIP(hipStreamSynchronize(stream));
  return NVBX_OK;
}
int nvbx_mapper::fetch_counters() {
  if (join_side()) return NVBX_E_DEVICE;
  NVBX_HIP(hipMemcpyAsync(h_counters, d.counters, C_NUM * 4, hipMemcpyDeviceToHost, stream));
  NVBX_HIP(hipMemcpyAsync(h_shc, d.shc, S_NUM * NSH * SH_STRIDE * 4, hipMemcpyDeviceToHost, stream));
  return wait_stream();
}

// log-odds of a probability, evaluated on the host in float (the oracle does the same with the same libm)
float nvbx::log_odds(float p) { return logf(p / (1.0f - p)); }

// T_L_C row-major 4x4 -> forward and inverse rigid transforms, fixed evaluation order (matches oracle rt_from_T)
Frame nvbx_mapper::make_frame(const float T[16], const nvbx_camera* cam, int32_t rows, int32_t cols, int32_t subsample) const {
  Frame f{};
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) f.R_LC[3 * i + j] = T[4 * i + j]; f.t_LC[i] = T[4 * i + 3]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) f.R_CL[3 * i + j] = f.R_LC[3 * j + i];
  for (int i = 0; i < 3; i++) {
    float s = f.R_CL[3 * i + 0] * f.t_LC[0];
    s = s + f.R_CL[3 * i + 1] * f.t_LC[1];
    s = s + f.R_CL[3 * i + 2] * f.t_LC[2];
    f.t_CL[i] = -s;
  }
  f.fu = cam->fu; f.fv = cam->fv; f.cu = cam->cu; f.cv = cam->cv; f.w = cam->width; f.h = cam->height;
  f.rows = rows; f.cols = cols;
  f.voxel_size = p.voxel_size; f.block_size = p.voxel_size * 8.0f;
  f.trunc = p.truncation_distance_vox * p.voxel_size;
  f.max_dist = p.max_integration_distance_m; f.max_weight = p.max_weight;
  f.weighting_mode = p.weighting_mode; f.interp_nearest = p.depth_interp_nearest;
  f.invalid_decay = p.invalid_depth_decay_factor;
  f.weighting_variant = p.tsdf_weighting_variant; f.skip_at_neg_trunc = p.tsdf_skip_at_negative_truncation; f.clamp_before_blend = p.tsdf_weight_clamp_before_blend;
  f.occlusion_thresh = p.color_occlusion_threshold_vox < 0.0f ? f.trunc : p.color_occlusion_threshold_vox * p.voxel_size;
  f.occupancy = p.projective_layer_type == 1 ? 1 : 0;
  f.lo_free = log_odds(p.free_region_occupancy_probability); f.lo_occupied = log_odds(p.occupied_region_occupancy_probability);
  f.lo_unobserved = log_odds(p.unobserved_region_occupancy_probability); f.occ_half_width = p.occupied_region_half_width_m;
  f.ws_type = p.workspace_bounds_type;
  for (int i = 0; i < 3; i++) { f.ws_min[i] = p.workspace_bounds_min_corner_m[i]; f.ws_max[i] = p.workspace_bounds_max_corner_m[i]; }
  f.subsample = subsample < 1 ? 1 : subsample;
  f.n_ray_rows = 0; f.n_ray_cols = 0;
  f.frame_id = frame_id;
  return f;
}

EsdfArgs nvbx_mapper::make_esdf_args() const {
  EsdfArgs c{};
  const float vs = p.voxel_size;
  c.kz_min = (int32_t)floorf(p.esdf_slice_min_height / vs);
  c.kz_max = (int32_t)floorf(p.esdf_slice_max_height / vs);
  c.kz_out = (int32_t)floorf(p.esdf_slice_height / vs);
  c.bz_lo = c.kz_min >> 3; c.bz_hi = c.kz_max >> 3; c.bz_out = c.kz_out >> 3; c.vz_out = c.kz_out & 7;
  const float r = p.esdf_max_distance_m / vs;
  c.max_sq = r * r;
  c.ri = (int32_t)floorf(r); if (c.ri > 63) c.ri = 63; if (c.ri < 1) c.ri = 1;
  c.rb = (c.ri + 7) / 8;
  c.site_dist_m = p.esdf_max_site_distance_vox * vs;
  c.min_weight = p.esdf_min_weight; c.voxel_size = vs; c.site_rule = p.projective_layer_type == 1 ? 2 : p.esdf_site_rule;
  c.epoch = esdf_epoch; c.mark_pass = mark_pass;
  c.rec = C_ESDF_UPD + 8 * (int)(esdf_epoch & 1); c.rec_next = C_ESDF_UPD + 8 * (int)((esdf_epoch + 1) & 1);
  c.self_reset = c.keep_list = pipelined_order ? 1 : 0;       // (see EsdfArgs)
  // ground-plane-relative band: only with a usable plane (pointing up) and a positive thickness; else the fixed heights
  c.plane_on = (p.esdf_use_ground_plane && p.esdf_ground_plane[2] > 1e-3f && p.slice_height_thickness_m > 0.0f) ? 1 : 0;
  for (int i = 0; i < 4; i++) c.pl[i] = p.esdf_ground_plane[i];
  c.above = p.slice_height_above_plane_m; c.thick = p.slice_height_thickness_m;
  return c;
}

// Parameter values the kernels' loop bounds and address arithmetic rely on (the reference CHECKs the same kind of thing in its
// setters and aborts; here the call fails with NVBX_E_INVALID).  Returns nullptr if fine, else what is wrong.
static const char* params_problem(const nvbx_mapper_params* p) {
  auto pos = [](float v) { return std::isfinite(v) && v > 0.0f; };
  if (!pos(p->voxel_size)) return "voxel_size must be > 0";
  if (!pos(p->max_integration_distance_m)) return "max_integration_distance_m must be > 0 (it bounds the view rays)";
  if (!pos(p->lidar_max_integration_distance_m)) return "lidar_max_integration_distance_m must be > 0 (it bounds the view rays)";
  if (p->max_integration_distance_m / (p->voxel_size * 8.0f) > 65536.0f || p->lidar_max_integration_distance_m / (p->voxel_size * 8.0f) > 65536.0f)
    return "integration distance exceeds 65536 blocks";
  if (!pos(p->truncation_distance_vox)) return "truncation_distance_vox must be > 0";
  if (!pos(p->max_weight)) return "max_weight must be > 0";
  if (p->projective_layer_type < 0 || p->projective_layer_type > 2) return "projective_layer_type must be 0 (TSDF), 1 (occupancy) or 2 (TSDF + freespace)";
  if (p->esdf_mode < 0 || p->esdf_mode > 1) return "esdf_mode must be 0 (2-D) or 1 (3-D)";
  if (p->tsdf_weighting_variant < 0 || p->tsdf_weighting_variant > 1 || p->esdf_propagation < 0 || p->esdf_propagation > 1 ||
      p->mesh_ambiguity_rule < 0 || p->mesh_ambiguity_rule > 2 || p->mesh_normal_rule < 0 || p->mesh_normal_rule > 1) return "open-choice switch out of range";
  if (p->esdf_propagation == 1 && p->esdf_mode == 1) return "esdf_propagation 1 (iterative) is defined for the 2-D slice only";
  if (p->sphere_tracing_max_steps < 0 || p->sphere_tracing_max_steps > (1 << 20)) return "sphere_tracing_max_steps out of range";
  if (p->projective_layer_type == 1) {
    auto prob = [](float v) { return v > 0.0f && v < 1.0f; };
    if (!prob(p->free_region_occupancy_probability) || !prob(p->occupied_region_occupancy_probability) || !prob(p->unobserved_region_occupancy_probability) ||
        !prob(p->free_region_decay_probability) || !prob(p->occupied_region_decay_probability)) return "occupancy probabilities must lie strictly between 0 and 1";
  }
  return nullptr;
}

// ------------------------------------------------------------------------------------------------ C-ABI: lifetime
extern "C" int nvbx_mapper_create(int device, void* hip_stream, const nvbx_mapper_params* params, int64_t block_capacity, nvbx_mapper** out) {
  if (!params || !out || (block_capacity != 0 && (block_capacity < 64 || block_capacity > (1ll << 24)))) {
    set_error("nvbx_mapper_create: invalid argument (null pointer, or block_capacity outside 64 .. 2^24 and not 0 = automatic)"); return NVBX_E_INVALID;
  }
  if (const char* why = params_problem(params)) { set_error(why); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(device));
  if (block_capacity == 0) {      // automatic: ~4 % of the free HBM (12.4 KiB per block), 2^16 .. 2^20 blocks (1 M blocks = 12 GiB of a 288 GB MI355X); grows on demand
    size_t free_b = 0, total_b = 0; NVBX_HIP(hipMemGetInfo(&free_b, &total_b));
    int64_t want = (int64_t)((double)free_b * 0.04 / 12700.0), cap = 1ll << 16;
    while (cap * 2 <= want && cap < (1ll << 20)) cap *= 2;
    block_capacity = cap;
  }
  nvbx_mapper* m = new nvbx_mapper();
  m->device = device; m->p = *params; m->capacity = block_capacity;
  // the pools double on demand up to this many blocks (nvbx_mapper_set_max_capacity; NVBX_MAX_BLOCKS in the environment)
  m->max_capacity = std::max<int64_t>(block_capacity, 1ll << 22);
  { const char* e = getenv("NVBX_MAX_BLOCKS"); if (e && atoll(e) > 0) m->max_capacity = std::max<int64_t>(block_capacity, std::min<int64_t>(atoll(e), 1ll << 24)); }
  if (hip_stream) { m->stream = (hipStream_t)hip_stream; m->own_stream = false; }
  else { hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking); if (e != hipSuccess) { set_error("hipStreamCreate", e); delete m; return NVBX_E_DEVICE; } m->own_stream = true; }
  nvbx::frames_register_stream(device, m->stream); m->stream_registered = true;      // (frames let go of without a fence wait for this stream too, frames.hip)
  // ESDF on a side stream beside colour integration: off by default, NVBX_SIDE_STREAM=1 enables (the
  // cross-stream hand-off costs ~10 us each way on this runtime, which eats most of the overlap)
  { const char* e = getenv("NVBX_SIDE_STREAM"); m->use_side = (e && e[0] == '1'); }
  { const char* e = getenv("NVBX_DEFER_EDT"); m->defer_edt = !(e && e[0] == '0'); }
  // colour deferral of a new mapper (nvbx_mapper_set_color_deferral overrides): NVBX_COLOR_DEFERRAL = 0 / 1 / 2 in the environment
  { const char* e = getenv("NVBX_COLOR_DEFERRAL"); if (e && e[0] >= '0' && e[0] <= '2' && !e[1]) { m->color_deferral = e[0] != '0'; m->color_staging = e[0] == '2'; } }
  if (m->use_side) {
    if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_main, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_side, hipEventDisableTiming) != hipSuccess) { set_error("side stream / events"); nvbx_mapper_destroy(m); return NVBX_E_DEVICE; }
  }
  int rc = alloc_all(m); if (rc) { nvbx_mapper_destroy(m); return rc; }
  rc = reset_map(m); if (rc) { nvbx_mapper_destroy(m); return rc; }
  hipError_t e = hipStreamSynchronize(m->stream);
  if (e != hipSuccess) { set_error("create sync", e); nvbx_mapper_destroy(m); return NVBX_E_DEVICE; }
  *out = m;
  return NVBX_OK;
}

extern "C" int nvbx_mapper_destroy(nvbx_mapper* m) {
  if (!m) return NVBX_OK;
  (void)hipSetDevice(m->device);
  if (m->side) (void)hipStreamSynchronize(m->side);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->ev_order) (void)hipEventDestroy(m->ev_order);
  if (m->ev_main) (void)hipEventDestroy(m->ev_main);
  if (m->ev_side) (void)hipEventDestroy(m->ev_side);
  if (m->side) (void)hipStreamDestroy(m->side);
  DMap& d = m->d;
  void* ptrs[] = {d.table, d.free_stack, d.counters, d.slot_flags, d.slot_index, d.slot_entry, d.slot_stamp, d.slot_consumed, d.slot_cam, d.tsdf, d.color, d.esdf,
                  m->view_list, d.lists, d.shc, m->export_idx, m->export_count, m->cleared_idx, d.site_bits, d.obs_bits, d.inside_bits,
                  m->table_spare, m->table_dirty, m->synth, m->view_class, m->view_grid_fine, m->color_cand, m->depth_pre, m->mask_zmin, m->apply_postab, m->esdf3_scratch, m->cc_scratch, m->dyn_scratch, d.freespace, m->lidar_tab, m->mesh_vert, m->mesh_nrm, m->mesh_col, m->mesh_tri, m->mesh_rec, m->staging};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  // (both streams are idle: whatever read a held-back colour frame has finished)
  (void)m->take_pending(); m->release_consumed_frames();
  if (m->stream_registered) nvbx::frames_forget_owner(m, m->device, m->stream);
  for (auto& s : m->spans) { if (s.a) (void)hipEventDestroy(s.a); if (s.b) (void)hipEventDestroy(s.b); }
  for (hipEvent_t e : m->event_pool) if (e) (void)hipEventDestroy(e);
  if (m->h_counters) (void)hipHostFree(m->h_counters);
  if (m->h_shc) (void)hipHostFree(m->h_shc);
  if (m->h_mirror) (void)hipHostFree(m->h_mirror);
  if (m->slice_pinned) (void)hipHostFree(m->slice_pinned);
  if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return NVBX_OK;
}

extern "C" int nvbx_mapper_set_params(nvbx_mapper* m, const nvbx_mapper_params* params) {
  if (!m || !params) return NVBX_E_INVALID;
  if (const char* why = params_problem(params)) { set_error(why); return NVBX_E_INVALID; }
  if (params->voxel_size != m->p.voxel_size) { set_error("voxel_size cannot change after creation"); return NVBX_E_INVALID; }
  if (params->projective_layer_type != m->p.projective_layer_type || params->esdf_mode != m->p.esdf_mode) {
    // what the voxels of the existing map MEAN would change under it (the reference fixes both at construction too)
    if (nvbx_num_blocks(m, F_TSDF | 0u) != 0 || nvbx_num_blocks(m, NVBX_LAYER_OCCUPANCY) != 0 || nvbx_num_blocks(m, F_ESDF) != 0) {
      set_error("projective_layer_type / esdf_mode can only change while the map is empty"); return NVBX_E_INVALID;
    }
  }
  if (m->join_side()) return NVBX_E_DEVICE;        // work enqueued under the old parameters (a held-back EDT) is launched first
  const bool trunc_changed = params->truncation_distance_vox != m->p.truncation_distance_vox;
  m->p = *params;
  if (trunc_changed && m->p.projective_layer_type != 1) {       // F_BAND is defined by the truncation distance: recompute it for every TSDF block
    NVBX_LAUNCH(m, k_recompute_band, dim3((unsigned)std::min<int64_t>(m->capacity, 2048)), dim3(512), m->d, m->p.truncation_distance_vox * m->p.voxel_size);
    NVBX_HIP(hipGetLastError());
  }
  return NVBX_OK;
}
#ifdef NVBX_CHECK_INVARIANTS
// Variant-only (tools/build_variant.sh inv "-DNVBX_CHECK_INVARIANTS"; not part of include/nvblox_hip.h): the violation counters of DESIGN.md 2.8's
// invariants as the kernels counted them -- out[0] = I1 (a TSDF-reading rider of launch 1 beside a running TSDF writer), out[1] = I3 (a colour worker
// handed a record whose slot does not name the block), out[2] = I4 (the marking pass took an entry of a slot without a layer), out[3] = writers
// still counted as running (must be 0 between launches), out[4] = I8 (host side: a colour-reading launch enqueued on a frame nobody holds).
// selftest != 0: first makes one reader meet a (pretended) writer, so a caller can see that the counters count.
__global__ void k_inv_selftest(DMap m) {
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&m.counters[C_INV_WRITERS], 1);
  __syncthreads();
  NVBX_INV_TSDF_READER(m);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicSub(&m.counters[C_INV_WRITERS], 1);
}
extern "C" int nvbx_debug_invariants(nvbx_mapper* m, int64_t out[5], int32_t selftest) {
  if (!m || !out) return NVBX_E_INVALID;
  if (selftest) { NVBX_LAUNCH(m, k_inv_selftest, dim3(1), dim3(64), m->d); NVBX_HIP(hipGetLastError()); }
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  out[0] = m->h_counters[C_INV_I1]; out[1] = m->h_counters[C_INV_I3]; out[2] = m->h_counters[C_INV_I4]; out[3] = m->h_counters[C_INV_WRITERS]; out[4] = m->inv_i8;
  return NVBX_OK;
}
#endif
extern "C" int nvbx_mapper_get_params(const nvbx_mapper* m, nvbx_mapper_params* out) {
  if (!m || !out) return NVBX_E_INVALID;
  *out = m->p; return NVBX_OK;
}
extern "C" int nvbx_synchronize(nvbx_mapper* m) {
  if (!m) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  return m->wait_stream();
}
extern "C" void nvbx_default_params(nvbx_mapper_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->voxel_size = 0.05f; p->max_integration_distance_m = 8.0f; p->truncation_distance_vox = 4.0f; p->max_weight = 5.0f;
  p->weighting_mode = 0; p->raycast_subsampling_factor = 4;
  p->esdf_min_weight = 0.1f; p->esdf_max_site_distance_vox = 2.0f; p->esdf_max_distance_m = 2.0f;
  p->esdf_slice_height = 0.09f; p->esdf_slice_min_height = 0.09f; p->esdf_slice_max_height = 0.65f;
  p->mesh_min_weight = 0.1f; p->mesh_weld_vertices = 1;
  p->sphere_tracing_subsampling = 4; p->sphere_tracing_max_steps = 100;
  p->sphere_tracing_max_ray_length_m = 15.0f; p->sphere_tracing_surface_eps_vox = 0.1f;
  p->tsdf_decay_factor = 0.95f; p->tsdf_decayed_weight_threshold = 0.001f;
  p->esdf_site_rule = 0; p->depth_interp_nearest = 0;
  p->lidar_max_integration_distance_m = 10.0f;
  p->lidar_linear_interpolation_max_allowable_difference_vox = 2.0f;
  p->lidar_nearest_interpolation_max_allowable_dist_to_ray_vox = 0.5f;
  p->invalid_depth_decay_factor = -1.0f;
  p->projective_layer_type = 0;
  p->free_region_occupancy_probability = 0.45f; p->occupied_region_occupancy_probability = 0.55f;
  p->unobserved_region_occupancy_probability = 0.5f; p->occupied_region_half_width_m = 0.1f;
  p->free_region_decay_probability = 0.55f; p->occupied_region_decay_probability = 0.30f;
  p->esdf_mode = 0;
  p->max_tsdf_distance_for_occupancy_m = 0.15f; p->max_unobserved_to_keep_consecutive_occupancy_ms = 200;
  p->min_duration_since_occupied_for_freespace_ms = 1000; p->min_consecutive_occupancy_duration_for_reset_ms = 2000;
  p->check_neighborhood = 1; p->initialize_to_high_confidence_freespace = 0;
  p->tsdf_weighting_variant = 0; p->tsdf_skip_at_negative_truncation = 0; p->tsdf_weight_clamp_before_blend = 0;
  p->color_occlusion_threshold_vox = -1.0f; p->esdf_propagation = 0; p->mesh_ambiguity_rule = 0; p->mesh_normal_rule = 0;
  p->decay_deallocate_decayed_blocks = 1; p->tsdf_set_free_distance_on_decayed = 0; p->tsdf_decayed_free_distance_vox = 4.0f; p->occupancy_decay_to_free = 0;
}
__global__ void k_selftest_arith(const float* a, const float* b, float* q, float* r, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    q[i] = NVBX_DIV(a[i], b[i]);
    r[i] = NVBX_SQRT(fabsf(a[i]));
  }
}
extern "C" int nvbx_selftest_arith(const float* a_dev, const float* b_dev, float* quot_dev, float* root_dev, int64_t n) {
  if (!a_dev || !b_dev || !quot_dev || !root_dev || n < 0) { set_error("nvbx_selftest_arith: invalid argument"); return NVBX_E_INVALID; }
  if (n > 0) k_selftest_arith<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256)>>>(a_dev, b_dev, quot_dev, root_dev, n);
  NVBX_HIP(hipGetLastError());
  NVBX_HIP(hipDeviceSynchronize());
  return NVBX_OK;
}
extern "C" int nvbx_get_stream(nvbx_mapper* m, void** hip_stream_out) {
  if (!m || !hip_stream_out) return NVBX_E_INVALID;
  *hip_stream_out = (void*)m->stream;
  return NVBX_OK;
}
// Two mappers on two streams (a MultiMapper whose foreground mapper runs beside the background mapper): `waiter`'s stream takes up work enqueued
// after this call only when everything enqueued on `producer`'s stream so far has finished.
extern "C" int nvbx_mapper_wait_for(nvbx_mapper* waiter, nvbx_mapper* producer) {
  if (!waiter || !producer) return NVBX_E_INVALID;
  if (waiter == producer || waiter->stream == producer->stream) return NVBX_OK;
  if (waiter->device != producer->device) { set_error("nvbx_mapper_wait_for: mappers on different devices"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(producer->device));
  if (!producer->ev_order) NVBX_HIP(hipEventCreateWithFlags(&producer->ev_order, hipEventDisableTiming));
  NVBX_HIP(hipEventRecord(producer->ev_order, producer->stream));
  NVBX_HIP(hipStreamWaitEvent(waiter->stream, producer->ev_order, 0));
  return NVBX_OK;
}
extern "C" int nvbx_flush(nvbx_mapper* m) {
  if (!m) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  return m->join_side();
}
extern "C" int nvbx_mapper_clear(nvbx_mapper* m) {
  if (!m) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  return reset_map(m);
}

// ------------------------------------------------------------------------------------------------ C-ABI: layer access
static bool single_layer(uint32_t layer) { return layer == F_TSDF || layer == F_COLOR || layer == F_ESDF || layer == F_MESH || layer == NVBX_LAYER_OCCUPANCY || layer == F_FREESPACE; }
// API layer id -> internal slot flag; 0 = this mapper cannot hold that layer (it reads as empty).  The projective layer of a
// mapper (TSDF or occupancy, Mapper's ProjectiveLayerType) lives in the same pool under the same internal flag.
static uint32_t internal_layer(const nvbx_mapper* m, uint32_t layer) {
  const bool occ = m->p.projective_layer_type == 1;
  if (layer == NVBX_LAYER_OCCUPANCY) return occ ? F_TSDF : 0u;
  if (layer == F_TSDF) return occ ? 0u : F_TSDF;
  if (layer == F_FREESPACE) return m->d.freespace ? F_FREESPACE : 0u;
  return layer;
}

static void sort_indices(nvbx_index3d* v, int64_t n) {
  std::sort(v, v + n, [](const nvbx_index3d& a, const nvbx_index3d& b) {
    if (a.x != b.x) return a.x < b.x; if (a.y != b.y) return a.y < b.y; return a.z < b.z; });
}

extern "C" int64_t nvbx_block_indices(nvbx_mapper* m, uint32_t layer, nvbx_index3d* out, int64_t capacity) {
  if (!m || !single_layer(layer)) return NVBX_E_INVALID;
  layer = internal_layer(m, layer);
  if (!layer) return 0;
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_LAUNCH(m, k_zero_tmp, dim3(1), dim3(1), m->d);
  NVBX_LAUNCH(m, k_collect_indices, dim3(256), dim3(256), m->d, layer, m->export_idx, (int32_t)m->capacity);
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  int64_t n = m->h_counters[C_TMP];
  if (n > m->capacity) n = m->capacity;
  const int64_t k = std::min<int64_t>(n, capacity);
  if (out && k > 0) {
    std::vector<nvbx_index3d> tmp((size_t)n);
    NVBX_HIP(hipMemcpy(tmp.data(), m->export_idx, (size_t)n * 12, hipMemcpyDeviceToHost));
    sort_indices(tmp.data(), n);
    memcpy(out, tmp.data(), (size_t)k * 12);
  }
  return n;
}
extern "C" int64_t nvbx_num_blocks(nvbx_mapper* m, uint32_t layer) { return nvbx_block_indices(m, layer, nullptr, 0); }

extern "C" int64_t nvbx_last_depth_view(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity) {
  if (!m) return NVBX_E_INVALID;
  if (m->last_view_frame == 0) return 0;
  if (m->join_side()) return NVBX_E_DEVICE;
  const uint32_t cam_mask = m->last_view_batch > 1 ? (1u << (m->last_view_batch - 1)) : 0u;
  NVBX_LAUNCH(m, k_viewlist_to_indices, dim3(64), dim3(256), m->d, (const int4*)m->view_list, C_VIEW_COUNT + (int)(m->last_view_frame & 3), cam_mask, m->export_idx, (int32_t)m->capacity);
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  int64_t n = m->h_counters[C_VIEW_COUNT + (m->last_view_frame & 3)]; if (n > m->capacity) n = m->capacity;
  std::vector<nvbx_index3d> tmp((size_t)std::max<int64_t>(n, 1));
  if (n > 0) NVBX_HIP(hipMemcpy(tmp.data(), m->export_idx, (size_t)n * 12, hipMemcpyDeviceToHost));
  tmp.resize((size_t)n);
  tmp.erase(std::remove_if(tmp.begin(), tmp.end(), [](const nvbx_index3d& i) { return i.x == INT32_MIN; }), tmp.end());   // deallocated since / not in the mask
  n = (int64_t)tmp.size();
  const int64_t k = std::min<int64_t>(n, capacity);
  if (out && k > 0) { sort_indices(tmp.data(), n); memcpy(out, tmp.data(), (size_t)k * 12); }
  return n;
}
extern "C" int64_t nvbx_last_color_view(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity) {
  if (!m) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_LAUNCH(m, k_shardlist_to_indices, dim3(64), dim3(256), m->d, (int)S_LIST_COLOR, m->export_idx, (int32_t)m->capacity, m->export_count);
  int32_t n32 = 0;
  NVBX_HIP(hipMemcpyAsync(&n32, m->export_count, 4, hipMemcpyDeviceToHost, m->stream));
  NVBX_HIP(hipStreamSynchronize(m->stream));
  const int64_t n = n32, k = std::min<int64_t>(n, capacity);
  if (out && k > 0) {
    std::vector<nvbx_index3d> tmp((size_t)n);
    NVBX_HIP(hipMemcpy(tmp.data(), m->export_idx, (size_t)n * 12, hipMemcpyDeviceToHost));
    sort_indices(tmp.data(), n);
    memcpy(out, tmp.data(), (size_t)k * 12);
  }
  return n;
}

static size_t ref_voxel_bytes(uint32_t layer) { return layer == F_ESDF ? sizeof(nvbx_esdf_voxel) : (layer == NVBX_LAYER_OCCUPANCY ? sizeof(nvbx_occupancy_voxel) : (layer == F_FREESPACE ? sizeof(nvbx_freespace_voxel) : 8)); }

extern "C" int nvbx_get_blocks(nvbx_mapper* m, uint32_t layer, const nvbx_index3d* idx, int64_t n, void* voxels_out, int32_t* found_out) {
  if (!m || !idx || !voxels_out || n < 0 || !(layer == F_TSDF || layer == F_COLOR || layer == F_ESDF || layer == NVBX_LAYER_OCCUPANCY || layer == F_FREESPACE)) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  const size_t bb = 512 * ref_voxel_bytes(layer);
  const uint32_t ilayer = internal_layer(m, layer);
  if (!ilayer) { if (found_out) memset(found_out, 0, (size_t)n * 4); return NVBX_OK; }
  const int64_t chunk = std::max<int64_t>(1, (int64_t)((m->staging_bytes - 65536) / (bb + 16)));
  for (int64_t o = 0; o < n; o += chunk) {
    const int64_t c = std::min(chunk, n - o);
    int32_t* d_idx = (int32_t*)m->staging; int32_t* d_found = d_idx + 3 * c;
    uint8_t* d_out = (uint8_t*)m->staging + (((size_t)c * 16 + 255) & ~(size_t)255);
    NVBX_HIP(hipMemcpyAsync(d_idx, idx + o, (size_t)c * 12, hipMemcpyHostToDevice, m->stream));
    NVBX_LAUNCH(m, k_gather_blocks, dim3((unsigned)c), dim3(512), m->d, ilayer, (int32_t)(layer == NVBX_LAYER_OCCUPANCY), d_idx, (int32_t)c, d_out, d_found);
    NVBX_HIP(hipMemcpyAsync((uint8_t*)voxels_out + (size_t)o * bb, d_out, (size_t)c * bb, hipMemcpyDeviceToHost, m->stream));
    if (found_out) NVBX_HIP(hipMemcpyAsync(found_out + o, d_found, (size_t)c * 4, hipMemcpyDeviceToHost, m->stream));
    NVBX_HIP(hipStreamSynchronize(m->stream));
  }
  return NVBX_OK;
}
extern "C" int nvbx_get_block(nvbx_mapper* m, uint32_t layer, nvbx_index3d idx, void* voxels_out) {
  int32_t found = 0;
  const int rc = nvbx_get_blocks(m, layer, &idx, 1, voxels_out, &found);
  if (rc) return rc;
  return found ? NVBX_OK : NVBX_E_NOTFOUND;
}
extern "C" int nvbx_set_blocks(nvbx_mapper* m, uint32_t layer, const nvbx_index3d* idx, int64_t n, const void* voxels_in) {
  if (!m || (n > 0 && (!voxels_in || !idx)) || n < 0 || !(layer == F_TSDF || layer == F_COLOR || layer == F_ESDF || layer == NVBX_LAYER_OCCUPANCY)) return NVBX_E_INVALID;
  const uint32_t ilayer = internal_layer(m, layer);
  if (!ilayer) { set_error("nvbx_set_blocks: this mapper's projective layer type does not hold that layer"); return NVBX_E_INVALID; }
  for (int64_t i = 0; i < n; i++) if (!nvbx_index_in_range(idx[i].x, idx[i].y, idx[i].z)) { set_error("nvbx_set_blocks: block index outside +-2^20"); return NVBX_E_INVALID; }
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->capacity < m->max_capacity) {          // explicit allocation (allocateBlockAtIndex, loadMap): room for all of it, plus the usual head-room
    if (m->fetch_counters()) return NVBX_E_DEVICE;
    m->h_mirror[0] = m->h_counters[C_FREE_TOP];
    const int rcg = m->maybe_grow(n); if (rcg) return rcg;
  }
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  const size_t bb = 512 * ref_voxel_bytes(layer);
  const EsdfArgs ea = m->make_esdf_args();
  const int64_t chunk = std::max<int64_t>(1, (int64_t)((m->staging_bytes - 65536) / (bb + 16)));
  for (int64_t o = 0; o < n; o += chunk) {
    const int64_t c = std::min(chunk, n - o);
    int32_t* d_idx = (int32_t*)m->staging;
    uint8_t* d_in = (uint8_t*)m->staging + (((size_t)c * 16 + 255) & ~(size_t)255);
    NVBX_HIP(hipMemcpyAsync(d_idx, idx + o, (size_t)c * 12, hipMemcpyHostToDevice, m->stream));
    NVBX_HIP(hipMemcpyAsync(d_in, (const uint8_t*)voxels_in + (size_t)o * bb, (size_t)c * bb, hipMemcpyHostToDevice, m->stream));
    NVBX_LAUNCH(m, k_scatter_blocks, dim3((unsigned)c), dim3(512), m->d, ilayer, (int32_t)(layer == NVBX_LAYER_OCCUPANCY), (const int32_t*)d_idx, (const uint8_t*)d_in, bb,
                (int32_t)m->mesh_list_live(), ea.bz_out, ea.vz_out, m->p.truncation_distance_vox * m->p.voxel_size);
    NVBX_HIP(hipStreamSynchronize(m->stream));
  }
  return NVBX_OK;
}
extern "C" int nvbx_set_block(nvbx_mapper* m, uint32_t layer, nvbx_index3d idx, const void* voxels_in) {
  return nvbx_set_blocks(m, layer, &idx, 1, voxels_in);
}

// ------------------------------------------------------------------------------------------------ map file
// Mapper::saveLayerCake / loadMap (nvblox_node.cpp:1668,1703).  The reference's .nvblx container is defined in the absent
// nvblox core; this is our own little-endian container of the same content: the TSDF, colour and ESDF layers as
// {Index3D, 512 reference voxel structs} per block.  Layout: MapFileHeader, then per layer MapLayerHeader,
// int32[n][3] indices (sorted), voxel structs.
namespace {
struct MapFileHeader { char magic[8]; uint32_t version; float voxel_size; uint32_t n_layers; uint32_t reserved; };
struct MapLayerHeader { uint32_t layer; uint32_t voxel_bytes; uint64_t n_blocks; };
const char kMapMagic[8] = {'N', 'V', 'B', 'X', 'M', 'A', 'P', '1'};
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
}  // namespace

// ---- .nvblx: the layer cake as an SQLite database ([U]: the reference's serializer (nvblox/serialization, absent) stores the layers in
// an SQLite file, saveLayerCake / loadMap of nvblox_node.cpp:1663-1703 take a *.nvblx path; the schema below is a guess at that layout,
// so files are readable with any sqlite3 tool but NOT verified against upstream's):
//   layers(layer_type TEXT PRIMARY KEY, voxel_size REAL, block_size REAL, voxel_bytes INTEGER, num_blocks INTEGER)
//   <layer_type>_blocks(index_x INTEGER, index_y INTEGER, index_z INTEGER, data BLOB, PRIMARY KEY(index_x, index_y, index_z))
// with layer_type in {tsdf_layer, color_layer, esdf_layer, occupancy_layer}; data = the block's 512 voxels as the reference's
// structs in z + 8y + 64x order.  libsqlite3 is loaded at run time (dlopen): no header / link dependency; without it (or for a path
// that does not end in .nvblx) the compact container of our own is written / read.
#include <dlfcn.h>
namespace {
struct Sqlite {
  void* lib = nullptr;
  int (*open)(const char*, void**) = nullptr; int (*close)(void*) = nullptr;
  int (*exec)(void*, const char*, int (*)(void*, int, char**, char**), void*, char**) = nullptr;
  int (*prepare)(void*, const char*, int, void**, const char**) = nullptr;
  int (*bind_int)(void*, int, int) = nullptr; int (*bind_double)(void*, int, double) = nullptr;
  int (*bind_text)(void*, int, const char*, int, void (*)(void*)) = nullptr; int (*bind_blob)(void*, int, const void*, int, void (*)(void*)) = nullptr;
  int (*step)(void*) = nullptr; int (*reset)(void*) = nullptr; int (*finalize)(void*) = nullptr;
  int (*column_int)(void*, int) = nullptr; double (*column_double)(void*, int) = nullptr; const void* (*column_blob)(void*, int) = nullptr;
  int (*column_bytes)(void*, int) = nullptr; const unsigned char* (*column_text)(void*, int) = nullptr;
  bool ok() const { return lib != nullptr; }
};
const Sqlite& sqlite() {
  static Sqlite s = [] {
    Sqlite q;
    for (const char* name : {"libsqlite3.so.0", "libsqlite3.so"}) { q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (q.lib) break; }
    if (!q.lib) return q;
    auto sym = [&](const char* n) { void* p = dlsym(q.lib, n); if (!p) q.lib = nullptr; return p; };
    void* lib = q.lib;
#define NVBX_SQL(field, name) *(void**)(&q.field) = sym(name)
    NVBX_SQL(open, "sqlite3_open"); NVBX_SQL(close, "sqlite3_close"); NVBX_SQL(exec, "sqlite3_exec"); NVBX_SQL(prepare, "sqlite3_prepare_v2");
    NVBX_SQL(bind_int, "sqlite3_bind_int"); NVBX_SQL(bind_double, "sqlite3_bind_double"); NVBX_SQL(bind_text, "sqlite3_bind_text"); NVBX_SQL(bind_blob, "sqlite3_bind_blob");
    NVBX_SQL(step, "sqlite3_step"); NVBX_SQL(reset, "sqlite3_reset"); NVBX_SQL(finalize, "sqlite3_finalize"); NVBX_SQL(column_int, "sqlite3_column_int");
    NVBX_SQL(column_double, "sqlite3_column_double"); NVBX_SQL(column_blob, "sqlite3_column_blob"); NVBX_SQL(column_bytes, "sqlite3_column_bytes");
    NVBX_SQL(column_text, "sqlite3_column_text");
#undef NVBX_SQL
    if (!q.lib) { dlclose(lib); }
    return q;
  }();
  return s;
}
constexpr int kSqlOk = 0, kSqlRow = 100, kSqlDone = 101;
struct LayerName { uint32_t layer; const char* name; };
const LayerName kLayerNames[] = {{F_TSDF, "tsdf_layer"}, {F_COLOR, "color_layer"}, {F_ESDF, "esdf_layer"}, {NVBX_LAYER_OCCUPANCY, "occupancy_layer"}};
bool ends_with(const char* s, const char* suffix) { const size_t a = strlen(s), b = strlen(suffix); return a >= b && strcmp(s + a - b, suffix) == 0; }
struct DbCloser { void* db; ~DbCloser() { if (db) sqlite().close(db); } };
struct StmtCloser { void* st; ~StmtCloser() { if (st) sqlite().finalize(st); } };
}  // namespace

static int save_map_nvblx(nvbx_mapper* m, const char* path) {
  const Sqlite& q = sqlite();
  remove(path);
  DbCloser db{nullptr};
  if (q.open(path, &db.db) != kSqlOk) { set_error("nvbx_save_map: cannot create the .nvblx (SQLite) file"); return NVBX_E_IO; }
  if (q.exec(db.db, "PRAGMA journal_mode=OFF; PRAGMA synchronous=OFF; BEGIN;"
                    "CREATE TABLE layers(layer_type TEXT PRIMARY KEY, voxel_size REAL, block_size REAL, voxel_bytes INTEGER, num_blocks INTEGER);", nullptr, nullptr, nullptr) != kSqlOk) {
    set_error("nvbx_save_map: SQLite schema"); return NVBX_E_IO; }
  const uint32_t layers[3] = {m->p.projective_layer_type == 1 ? NVBX_LAYER_OCCUPANCY : F_TSDF, F_COLOR, F_ESDF};
  for (uint32_t layer : layers) {
    const char* name = nullptr; for (const LayerName& ln : kLayerNames) if (ln.layer == layer) name = ln.name;
    const int64_t n = nvbx_num_blocks(m, layer);
    if (n < 0) return (int)n;
    std::vector<nvbx_index3d> idx((size_t)std::max<int64_t>(n, 1));
    if (n > 0 && nvbx_block_indices(m, layer, idx.data(), n) < 0) return NVBX_E_DEVICE;
    const size_t bb = 512 * ref_voxel_bytes(layer);
    char sql[256];
    snprintf(sql, sizeof(sql), "CREATE TABLE %s_blocks(index_x INTEGER, index_y INTEGER, index_z INTEGER, data BLOB, PRIMARY KEY(index_x, index_y, index_z));", name);
    if (q.exec(db.db, sql, nullptr, nullptr, nullptr) != kSqlOk) { set_error("nvbx_save_map: SQLite create table"); return NVBX_E_IO; }
    {
      StmtCloser st{nullptr};
      if (q.prepare(db.db, "INSERT INTO layers VALUES(?, ?, ?, ?, ?);", -1, &st.st, nullptr) != kSqlOk) { set_error("nvbx_save_map: SQLite prepare"); return NVBX_E_IO; }
      q.bind_text(st.st, 1, name, -1, nullptr); q.bind_double(st.st, 2, (double)m->p.voxel_size); q.bind_double(st.st, 3, (double)(m->p.voxel_size * 8.0f));
      q.bind_int(st.st, 4, (int)ref_voxel_bytes(layer)); q.bind_int(st.st, 5, (int)n);
      if (q.step(st.st) != kSqlDone) { set_error("nvbx_save_map: SQLite insert"); return NVBX_E_IO; }
    }
    snprintf(sql, sizeof(sql), "INSERT INTO %s_blocks VALUES(?, ?, ?, ?);", name);
    StmtCloser st{nullptr};
    if (q.prepare(db.db, sql, -1, &st.st, nullptr) != kSqlOk) { set_error("nvbx_save_map: SQLite prepare"); return NVBX_E_IO; }
    const int64_t chunk = 4096;
    std::vector<uint8_t> buf((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), chunk) * bb);
    for (int64_t o = 0; o < n; o += chunk) {
      const int64_t c = std::min(chunk, n - o);
      const int rc = nvbx_get_blocks(m, layer, idx.data() + o, c, buf.data(), nullptr);
      if (rc) return rc;
      for (int64_t i = 0; i < c; i++) {
        q.reset(st.st);
        q.bind_int(st.st, 1, idx[(size_t)(o + i)].x); q.bind_int(st.st, 2, idx[(size_t)(o + i)].y); q.bind_int(st.st, 3, idx[(size_t)(o + i)].z);
        q.bind_blob(st.st, 4, buf.data() + (size_t)i * bb, (int)bb, nullptr);          // (static: the buffer outlives the step)
        if (q.step(st.st) != kSqlDone) { set_error("nvbx_save_map: SQLite insert block"); return NVBX_E_IO; }
      }
    }
  }
  if (q.exec(db.db, "COMMIT;", nullptr, nullptr, nullptr) != kSqlOk) { set_error("nvbx_save_map: SQLite commit"); return NVBX_E_IO; }
  return NVBX_OK;
}

static int load_map_nvblx(nvbx_mapper* m, const char* path) {
  const Sqlite& q = sqlite();
  DbCloser db{nullptr};
  if (q.open(path, &db.db) != kSqlOk) { set_error("nvbx_load_map: cannot open the .nvblx (SQLite) file"); return NVBX_E_IO; }
  // validate before the current map is touched: the layer table, voxel size, voxel struct sizes, block counts, blob sizes
  struct L { uint32_t layer; std::string name; int64_t n; size_t bb; };
  std::vector<L> found;
  {
    StmtCloser st{nullptr};
    if (q.prepare(db.db, "SELECT layer_type, voxel_size, voxel_bytes, num_blocks FROM layers;", -1, &st.st, nullptr) != kSqlOk) { set_error("nvbx_load_map: not an .nvblx layer cake"); return NVBX_E_IO; }
    int rc;
    while ((rc = q.step(st.st)) == kSqlRow) {
      const char* nm = (const char*)q.column_text(st.st, 0);
      uint32_t layer = 0; for (const LayerName& ln : kLayerNames) if (nm && !strcmp(nm, ln.name)) layer = ln.layer;
      const double vs = q.column_double(st.st, 1); const int vb = q.column_int(st.st, 2); const int64_t n = q.column_int(st.st, 3);
      if (!layer || vb != (int)ref_voxel_bytes(layer) || n < 0) { set_error("nvbx_load_map: unknown layer record"); return NVBX_E_IO; }
      if (fabs(vs - (double)m->p.voxel_size) > 1e-6 * m->p.voxel_size) { set_error("nvbx_load_map: voxel size of the file differs from the mapper's"); return NVBX_E_INVALID; }
      if (n > 0 && !internal_layer(m, layer)) { set_error("nvbx_load_map: the file holds a layer this mapper's projective layer type cannot"); return NVBX_E_IO; }
      if (n > (1ll << 24)) { set_error("nvbx_load_map: implausible block count"); return NVBX_E_CAPACITY; }
      found.push_back({layer, nm, n, 512 * ref_voxel_bytes(layer)});
    }
    if (rc != kSqlDone || found.empty()) { set_error("nvbx_load_map: not an .nvblx layer cake"); return NVBX_E_IO; }
  }
  for (const L& l : found) {
    char sql[256]; snprintf(sql, sizeof(sql), "SELECT COUNT(*), MIN(LENGTH(data)), MAX(LENGTH(data)) FROM %s_blocks;", l.name.c_str());
    StmtCloser st{nullptr};
    if (q.prepare(db.db, sql, -1, &st.st, nullptr) != kSqlOk || q.step(st.st) != kSqlRow) { set_error("nvbx_load_map: block table missing"); return NVBX_E_IO; }
    const int64_t cnt = q.column_int(st.st, 0);
    if (cnt != l.n || (cnt > 0 && (q.column_int(st.st, 1) != (int)l.bb || q.column_int(st.st, 2) != (int)l.bb))) { set_error("nvbx_load_map: block table does not match its layer record"); return NVBX_E_IO; }
  }
  int rc = nvbx_mapper_clear(m);
  if (rc) return rc;
  for (const L& l : found) {
    char sql[256]; snprintf(sql, sizeof(sql), "SELECT index_x, index_y, index_z, data FROM %s_blocks;", l.name.c_str());
    StmtCloser st{nullptr};
    if (q.prepare(db.db, sql, -1, &st.st, nullptr) != kSqlOk) { set_error("nvbx_load_map: SQLite prepare"); return NVBX_E_IO; }
    const int64_t chunk = 2048;
    std::vector<nvbx_index3d> idx; std::vector<uint8_t> buf;
    idx.reserve((size_t)chunk); buf.reserve((size_t)chunk * l.bb);
    int s;
    for (;;) {
      s = q.step(st.st);
      if (s == kSqlRow) {
        idx.push_back({q.column_int(st.st, 0), q.column_int(st.st, 1), q.column_int(st.st, 2)});
        const uint8_t* blob = (const uint8_t*)q.column_blob(st.st, 3);
        buf.insert(buf.end(), blob, blob + l.bb);
      }
      if ((s != kSqlRow || (int64_t)idx.size() == chunk) && !idx.empty()) {
        rc = nvbx_set_blocks(m, l.layer, idx.data(), (int64_t)idx.size(), buf.data());
        if (rc) return rc;
        idx.clear(); buf.clear();
      }
      if (s != kSqlRow) break;
    }
    if (s != kSqlDone) { set_error("nvbx_load_map: SQLite read"); return NVBX_E_IO; }
  }
  return NVBX_OK;
}

extern "C" int nvbx_save_map(nvbx_mapper* m, const char* path) {
  if (!m || !path) { set_error("nvbx_save_map: invalid argument"); return NVBX_E_INVALID; }
  if (ends_with(path, ".nvblx") && sqlite().ok()) return save_map_nvblx(m, path);
  FileCloser fc{fopen(path, "wb")};
  if (!fc.f) { set_error("nvbx_save_map: cannot open file for writing"); return NVBX_E_IO; }
  const uint32_t layers[3] = {m->p.projective_layer_type == 1 ? NVBX_LAYER_OCCUPANCY : F_TSDF, F_COLOR, F_ESDF};
  MapFileHeader h{}; memcpy(h.magic, kMapMagic, 8); h.version = 1; h.voxel_size = m->p.voxel_size; h.n_layers = 3;
  if (fwrite(&h, sizeof(h), 1, fc.f) != 1) { set_error("nvbx_save_map: write failed"); return NVBX_E_IO; }
  for (uint32_t layer : layers) {
    const int64_t n = nvbx_num_blocks(m, layer);
    if (n < 0) return (int)n;
    std::vector<nvbx_index3d> idx((size_t)std::max<int64_t>(n, 1));
    if (n > 0 && nvbx_block_indices(m, layer, idx.data(), n) < 0) return NVBX_E_DEVICE;
    const size_t bb = 512 * ref_voxel_bytes(layer);
    MapLayerHeader lh{layer, (uint32_t)ref_voxel_bytes(layer), (uint64_t)n};
    if (fwrite(&lh, sizeof(lh), 1, fc.f) != 1) { set_error("nvbx_save_map: write failed"); return NVBX_E_IO; }
    if (n > 0 && fwrite(idx.data(), 12, (size_t)n, fc.f) != (size_t)n) { set_error("nvbx_save_map: write failed"); return NVBX_E_IO; }
    const int64_t chunk = 4096;                      // blocks per round trip (16-40 MiB of host memory)
    std::vector<uint8_t> buf((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), chunk) * bb);
    for (int64_t o = 0; o < n; o += chunk) {
      const int64_t c = std::min(chunk, n - o);
      const int rc = nvbx_get_blocks(m, layer, idx.data() + o, c, buf.data(), nullptr);
      if (rc) return rc;
      if (fwrite(buf.data(), bb, (size_t)c, fc.f) != (size_t)c) { set_error("nvbx_save_map: write failed"); return NVBX_E_IO; }
    }
  }
  return NVBX_OK;
}

extern "C" int nvbx_load_map(nvbx_mapper* m, const char* path) {
  if (!m || !path) { set_error("nvbx_load_map: invalid argument"); return NVBX_E_INVALID; }
  FileCloser fc{fopen(path, "rb")};
  if (!fc.f) { set_error("nvbx_load_map: cannot open file"); return NVBX_E_IO; }
  { char magic[16] = {0};                             // an SQLite database = an .nvblx layer cake
    if (fread(magic, 1, 16, fc.f) == 16 && !memcmp(magic, "SQLite format 3", 16)) {
      if (!sqlite().ok()) { set_error("nvbx_load_map: the file is an SQLite .nvblx but libsqlite3 cannot be loaded"); return NVBX_E_IO; }
      fclose(fc.f); fc.f = nullptr;
      return load_map_nvblx(m, path);
    }
    fseek(fc.f, 0, SEEK_SET); }
  MapFileHeader h{};
  if (fread(&h, sizeof(h), 1, fc.f) != 1 || memcmp(h.magic, kMapMagic, 8) != 0 || h.version != 1) { set_error("nvbx_load_map: not a libnvblox_hip map file"); return NVBX_E_IO; }
  if (fabsf(h.voxel_size - m->p.voxel_size) > 1e-6f * m->p.voxel_size) { set_error("nvbx_load_map: voxel size of the file differs from the mapper's"); return NVBX_E_INVALID; }
  if (h.n_layers > 16) { set_error("nvbx_load_map: implausible layer count"); return NVBX_E_IO; }
  // validate the whole file before the current map is touched
  struct Section { MapLayerHeader lh; long idx_off, vox_off; };
  std::vector<Section> sections;
  for (uint32_t l = 0; l < h.n_layers; l++) {
    Section sc{};
    if (fread(&sc.lh, sizeof(sc.lh), 1, fc.f) != 1) { set_error("nvbx_load_map: truncated file"); return NVBX_E_IO; }
    if (!(sc.lh.layer == F_TSDF || sc.lh.layer == F_COLOR || sc.lh.layer == F_ESDF || sc.lh.layer == NVBX_LAYER_OCCUPANCY) ||
        sc.lh.voxel_bytes != ref_voxel_bytes(sc.lh.layer) || (sc.lh.n_blocks > 0 && !internal_layer(m, sc.lh.layer))) {
      set_error("nvbx_load_map: unknown layer record"); return NVBX_E_IO; }
    if (sc.lh.n_blocks > (uint64_t)m->capacity) { set_error("nvbx_load_map: map has more blocks than the mapper's block capacity"); return NVBX_E_CAPACITY; }
    sc.idx_off = ftell(fc.f); sc.vox_off = sc.idx_off + (long)(sc.lh.n_blocks * 12);
    if (fseek(fc.f, sc.vox_off + (long)(sc.lh.n_blocks * 512 * sc.lh.voxel_bytes), SEEK_SET) != 0) { set_error("nvbx_load_map: truncated file"); return NVBX_E_IO; }
    sections.push_back(sc);
  }
  { const long end = ftell(fc.f); fseek(fc.f, 0, SEEK_END); if (ftell(fc.f) < end) { set_error("nvbx_load_map: truncated file"); return NVBX_E_IO; } }
  int rc = nvbx_mapper_clear(m);
  if (rc) return rc;
  for (const Section& sc : sections) {
    const int64_t n = (int64_t)sc.lh.n_blocks;
    const size_t bb = 512 * (size_t)sc.lh.voxel_bytes;
    std::vector<nvbx_index3d> idx((size_t)std::max<int64_t>(n, 1));
    fseek(fc.f, sc.idx_off, SEEK_SET);
    if (n > 0 && fread(idx.data(), 12, (size_t)n, fc.f) != (size_t)n) { set_error("nvbx_load_map: read failed"); return NVBX_E_IO; }
    const int64_t chunk = 4096;
    std::vector<uint8_t> buf((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), chunk) * bb);
    for (int64_t o = 0; o < n; o += chunk) {
      const int64_t c = std::min(chunk, n - o);
      if (fread(buf.data(), bb, (size_t)c, fc.f) != (size_t)c) { set_error("nvbx_load_map: read failed"); return NVBX_E_IO; }
      rc = nvbx_set_blocks(m, sc.lh.layer, idx.data() + o, c, buf.data());
      if (rc) return rc;
    }
  }
  return NVBX_OK;
}

extern "C" int nvbx_get_counters(nvbx_mapper* m, nvbx_counters* out) {
  if (!m || !out) return NVBX_E_INVALID;
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  const int32_t* c = m->h_counters;
  memset(out, 0, sizeof(*out));
  out->blocks_allocated = (int64_t)m->capacity - (int64_t)c[C_FREE_TOP];       // every slot is free or live
  out->tsdf_blocks_in_view = m->last_view_frame ? c[C_VIEW_COUNT + (m->last_view_frame & 3)] : 0;
  out->color_blocks_updated = m->shc_sum(S_LIST_COLOR, 0);
  const int epar = (int)((m->esdf_epoch + 1) & 1);                    // record of the last finished update (epoch - 1)
  out->esdf_columns_marked = m->esdf_epoch ? m->shc_sum(S_ESDF_REC + epar, 4) : 0;
  out->esdf_blocks_swept = m->esdf_epoch ? m->shc_sum(S_ESDF_REC + epar, 5) : 0;
  out->esdf_window_voxels = m->esdf_epoch ? c[C_ESDF_UPD + 8 * epar + 6] : 0;
  if (m->p.esdf_mode == 1) { out->esdf_columns_marked = m->esdf3_blocks_marked; out->esdf_blocks_swept = m->esdf3_window_voxels / 512; out->esdf_window_voxels = m->esdf3_window_voxels; }
  const int mpar = (int)((m->mesh_epoch + 1) & 1);                    // record of the last finished mesh update
  out->mesh_blocks_updated = m->mesh_epoch ? m->shc_sum(S_MESH_REC + mpar, 0) : 0;
  out->mesh_vertices = m->mesh_epoch ? m->shc_sum(S_MESH_REC + mpar, 2) : 0;
  out->mesh_triangles = m->mesh_epoch ? m->shc_sum(S_MESH_REC + mpar, 3) : 0;
  out->capacity_overflow = c[C_OVERFLOW];
  out->lidar_blocks_beam_centric = m->shc_sum(S_LIDAR_SPARSE, 0);
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ per-kernel timing
hipEvent_t nvbx_mapper::get_event() {
  if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
void nvbx_mapper::span_begin(const char* name, hipStream_t st) {
  Span s{name, get_event(), get_event()};
  (void)hipEventRecord(s.a, st);
  spans.push_back(s);
}
void nvbx_mapper::span_end(hipStream_t st) { (void)hipEventRecord(spans.back().b, st); }

__global__ void k_reset_esdf_dirty_list(DMap m) { if (threadIdx.x < NSH) *shc_at(m, S_LIST_ESDF_DIRTY, threadIdx.x, 0) = 0; }
int nvbx_mapper::reset_consumed_list() {
  if (premark_consumed) { NVBX_LAUNCH(this, k_reset_esdf_dirty_list, dim3(1), dim3(64), d); premark_consumed = false; }
  return NVBX_OK;
}
int nvbx_mapper::join_side() {
  enqueue_seq++;                     // (an entry point runs: host copies of device counters are stale from here on, esdf.hip nvbx_esdf_slice_to_image)
  zc_valid = false;                  // (whatever follows may change the TSDF: the kept zero-crossing list is dropped)
  // every entry point passes here before its first HIP call: make this mapper's device current (hosts with one mapper per GPU in
  // one process); a thread-local read when it already is
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != device) NVBX_HIP(hipSetDevice(device)); }
  if (!replaying && !pipelined_order && replay_deferred()) return NVBX_E_DEVICE;      // held-back integrateColor / updateEsdf: carried out first, in call order
  if (flush_edt()) return NVBX_E_DEVICE;
  if (flush_import()) return NVBX_E_DEVICE;
  main_dirty = true;
  if (side_pending) { NVBX_HIP(hipStreamWaitEvent(stream, ev_side, 0)); side_pending = false; }
  return NVBX_OK;
}
int nvbx_mapper::join_side_keeping_held() {
  const bool e = edt_pending, i = import_pending, u = esdf_update_pending, c = color_pending.on;
  edt_pending = false; import_pending = false; esdf_update_pending = false; color_pending.on = false;
  const int rc = join_side();
  edt_pending = e; import_pending = i; esdf_update_pending = u; color_pending.on = c;
  return rc;
}
// the held-back calls of colour deferral, carried out as they would have been at call time
int nvbx_mapper::replay_deferred() {
  if (!color_pending.on && !esdf_update_pending) return NVBX_OK;
  replaying = true;
  int rc = NVBX_OK;
  if (replay_pair_applies()) { rc = replay_pair(); replaying = false; return rc == NVBX_OK ? NVBX_OK : NVBX_E_DEVICE; }
  if (color_pending.on) {
    const ColorPending c = take_pending();
    if (c.n > 1) rc = nvbx_integrate_color_batch(this, c.n, reinterpret_cast<const uint8_t* const*>(c.imgs), c.rows, c.cols, c.T, c.cams);
    else rc = c.kind == 0 ? nvbx_integrate_color(this, (const uint8_t*)c.imgs[0], c.rows, c.cols, c.T, &c.cams[0])
                          : nvbx_integrate_color_bgra8(this, (const uint8_t*)c.imgs[0], c.rows, c.cols, c.T, &c.cams[0]);
  }
  if (rc == NVBX_OK && esdf_update_pending) { esdf_update_pending = false; rc = nvbx_update_esdf(this); }
  esdf_update_pending = false;
  replaying = false;
  release_consumed_frames();
  return rc == NVBX_OK ? NVBX_OK : NVBX_E_DEVICE;
}
extern "C" int nvbx_mapper_set_color_deferral(nvbx_mapper* m, int32_t enable) {
  if (!m) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;          // (anything held back under the old setting is carried out)
  if (enable < 0 || enable > 2) { set_error("nvbx_mapper_set_color_deferral: 0 = off, 1 = on (the caller keeps the image valid), 2 = on with a staged copy"); return NVBX_E_INVALID; }
  m->color_deferral = enable != 0; m->color_staging = enable == 2;
  return NVBX_OK;
}
int nvbx_mapper::mark_main() {
  if (use_side) { NVBX_HIP(hipEventRecord(ev_main, stream)); main_dirty = false; }
  return NVBX_OK;
}

extern "C" int nvbx_set_profiling(nvbx_mapper* m, int32_t enable) {
  if (!m) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_HIP(hipStreamSynchronize(m->stream));
  for (auto& s : m->spans) { m->event_pool.push_back(s.a); m->event_pool.push_back(s.b); }
  m->spans.clear();
  m->profiling = enable != 0;
  return NVBX_OK;
}

// JSON object {"kernel": {"count": n, "total_ms": t}, ...} of every launch since nvbx_set_profiling(m, 1).
extern "C" int nvbx_get_profile(nvbx_mapper* m, char* json_out, int64_t capacity) {
  if (!m || !json_out || capacity < 4) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_HIP(hipStreamSynchronize(m->stream));
  struct Acc { const char* name; int64_t n; double ms; double max_ms; };
  std::vector<Acc> acc;
  for (auto& s : m->spans) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
    bool hit = false;
    for (auto& a : acc) if (!strcmp(a.name, s.name)) { a.n++; a.ms += ms; if (ms > a.max_ms) a.max_ms = ms; hit = true; break; }
    if (!hit) acc.push_back({s.name, 1, ms, ms});
  }
  // what a hipEvent pair adds to the span of ONE launch: pairs with nothing between them, on the same (now idle) stream
  {
    const int kPairs = 32; double ms_sum = 0.0; int n_ok = 0;
    std::vector<hipEvent_t> ev;
    for (int i = 0; i < 2 * kPairs; i++) { ev.push_back(m->get_event()); (void)hipEventRecord(ev.back(), m->stream); }
    NVBX_HIP(hipStreamSynchronize(m->stream));
    for (int i = 0; i < kPairs; i++) { float ms = 0.0f; if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) { ms_sum += ms; n_ok++; } }
    for (hipEvent_t e : ev) m->event_pool.push_back(e);
    if (n_ok) acc.push_back({"_empty_event_pair", n_ok, ms_sum, 0.0});
  }
  std::string out = "{";
  for (size_t i = 0; i < acc.size(); i++) {
    char buf[256];
    std::string nm = acc[i].name;
    for (char& c : nm) if (c == '(' || c == ')' ) c = ' ';
    snprintf(buf, sizeof(buf), "%s\"%s\": {\"count\": %lld, \"total_ms\": %.6f, \"max_ms\": %.6f}", i ? ", " : "", nm.c_str(), (long long)acc[i].n, acc[i].ms, acc[i].max_ms);
    out += buf;
  }
  out += "}";
  if ((int64_t)out.size() + 1 > capacity) return NVBX_E_CAPACITY;
  memcpy(json_out, out.c_str(), out.size() + 1);
  return NVBX_OK;
}
