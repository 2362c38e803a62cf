// tsdf.hip -- MultiMapper::integrateDepth on MI355X: block marking (view calculation) + projective TSDF update.
//
// Two launches per depth frame, no host round trip in between:
//   k_mark_view      one wavefront per 8x8 tile of the sub-sampled ray grid.  Phase 1: each lane walks its ray through
//                    the block grid (Amanatides-Woo) and drops the block keys into a 4 KiB LDS set (rays of one tile
//                    share almost all their blocks) -- no HBM access inside the walk.  Phase 2: the set is compacted
//                    (ballot + popcount) and ONE key per lane goes to HBM: CAS insert-if-absent into the hash
//                    (device-side allocation from the slot stack), per-entry frame stamp, and a wave-aggregated append
//                    of the pool slot to the frame's view list (exactly once per block and frame).
//   k_integrate_tsdf one 512-thread workgroup (8 wave64) per 8^3 block, grid-striding over the device-resident view
//                    list of {slot, Index3D} records; lane = voxel in z + 8y + 64x order, so every wave reads/writes
//                    512 contiguous bytes.
// Reference semantics restated: [U] ViewCalculator::getBlocksInImageViewRaycast and ProjectiveTsdfIntegrator
// (call site nvblox_ros/src/lib/nvblox_node.cpp:1062; knobs mapper_initialization.cpp:264-358).
#include <algorithm>
#include "nvbx_mapper.h"

using namespace nvbx;

constexpr int LSET = 512;   // LDS dedup set entries per wave-tile (8 B each)

__device__ inline void unpack_key(u64 key, int32_t* x, int32_t* y, int32_t* z) {
  *x = (int32_t)(key & 0x1FFFFFull) - (1 << 20);
  *y = (int32_t)((key >> 21) & 0x1FFFFFull) - (1 << 20);
  *z = (int32_t)((key >> 42) & 0x1FFFFFull) - (1 << 20);
}

// One block key -> HBM: insert-if-absent, stamp the entry with this frame, and report whether THIS call was the first
// of the frame to do so (the caller then appends {slot, x, y, z} to the view list exactly once).  The common case -- the
// block exists and a neighbouring tile has stamped it already -- is ONE 16-B load: key, slot and stamp arrive together.
__device__ inline bool mark_block(const DMap& m, u64 key, uint32_t frame_id, int4* rec_out) {
  int32_t x, y, z; unpack_key(key, &x, &y, &z);
  uint32_t h = table_pos(m, x, y, z);
  uint32_t slot = SLOT_INVALID;
  bool found = false;
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&m.table[h]);
    const u64 k = ((u64)e.y << 32) | (u64)e.x;
    if (k == key) { if (e.w == frame_id) return false; slot = e.z; found = true; break; }
    if (k == KEY_EMPTY) break;           // (may be a stale EMPTY: hash_insert's CAS is the truth)
    h = (h + 1) & m.mask;
  }
  if (!found) {
    bool is_new;
    const int32_t hi = hash_insert(m, x, y, z, F_TSDF, &is_new);
    if (hi < 0) return false;
    h = (uint32_t)hi;
  }
  if (atomicExch(&m.table[h].stamp, frame_id) == frame_id) return false;
  while (slot == SLOT_INVALID) slot = ld_slot_acquire(&m.table[h]);     // the inserting lane publishes right after its CAS
  *rec_out = make_int4((int32_t)slot, x, y, z);
  return true;
}

// wave-aggregated append of this lane's record (if `first`) to the frame's view list: one returning atomic per wave
__device__ inline void view_append(int32_t* cnt, int4* view_list, int32_t list_cap, bool first, int4 rec, int lane) {
  const u64 mask = __ballot(first);
  if (!mask) return;
  int32_t base = 0;
  const int leader = __ffsll((long long)mask) - 1;
  if (lane == leader) base = atomicAdd(cnt, (int32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (first) {
    const int32_t pos = base + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (pos < list_cap) view_list[pos] = rec;
  }
}

template <typename Img>
__global__ __launch_bounds__(64) void k_mark_view(DMap m, Frame f, Img depth, int4* view_list, int32_t list_cap) {
  __shared__ u64 lset[LSET];
  __shared__ u64 lkeys[LSET];
  const int lane = threadIdx.x;
  for (int i = lane; i < LSET; i += 64) lset[i] = KEY_EMPTY;
  if (blockIdx.x == 0 && lane == 0) m.counters[C_VIEW_COUNT + ((f.frame_id + 1) & 3)] = 0;   // next frame's counter
  __syncthreads();

  const int tiles_x = (f.n_ray_cols + 7) >> 3;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int ri = ty * 8 + (lane >> 3), ci = tx * 8 + (lane & 7);
  bool active = ri < f.n_ray_rows && ci < f.n_ray_cols;

  int32_t cur[3] = {0, 0, 0}, step[3] = {0, 0, 0}, nsteps = -1;
  float tmax[3] = {0, 0, 0}, tdelta[3] = {0, 0, 0};
  if (active) {
    int prow = ri * f.subsample; if (prow >= f.rows) prow = f.rows - 1;
    int pcol = ci * f.subsample; if (pcol >= f.cols) pcol = f.cols - 1;
    const float d = depth((int64_t)prow * f.cols + pcol);
    if (!(d > 0.0f)) active = false;
    else {
      float de = d + f.trunc;
      if (f.max_dist > 0.0f && de > f.max_dist) de = f.max_dist;
      const float rx = (((float)pcol + 0.5f) - f.cu) / f.fu;
      const float ry = (((float)prow + 0.5f) - f.cv) / f.fv;
      float pl[3];
      apply_rt(f.R_LC, f.t_LC, de * rx, de * ry, de, pl);
      nsteps = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float s = f.t_LC[a] / f.block_size, t = pl[a] / f.block_size;
        cur[a] = (int32_t)floorf(s);
        const int32_t end = (int32_t)floorf(t);
        const int32_t dd = end - cur[a]; nsteps += dd < 0 ? -dd : dd;
        const float ray = t - s;
        step[a] = ray > 0.0f ? 1 : (ray < 0.0f ? -1 : 0);
        const float corrected = step[a] > 0 ? 1.0f : 0.0f;
        const float dist_to_boundary = corrected - (s - (float)cur[a]);
        if (fabsf(ray) < 1e-9f) { tmax[a] = 2.0f; tdelta[a] = 2.0f; }
        else { tmax[a] = dist_to_boundary / ray; tdelta[a] = (float)step[a] / ray; }
      }
    }
  }
  int32_t* cnt = &m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  // ---- phase 1: walk.  Keys go to the LDS set only; HBM is touched here only if the set overflows (never at
  //      640x480 / 8 m: a tile's ray bundle crosses < 100 blocks).
  for (int32_t k = 0; k <= nsteps; k++) {
    const u64 key = pack_key(cur[0], cur[1], cur[2]);
    bool spill = true;
    const uint32_t lh = (index_hash(cur[0], cur[1], cur[2]) * 2654435761u) >> 23;   // 9 bits
#pragma unroll 1
    for (int p = 0; p < 16; p++) {
      const u64 old = atomicCAS(&lset[(lh + p) & (LSET - 1)], KEY_EMPTY, key);
      if (old == KEY_EMPTY || old == key) { spill = false; break; }
    }
    if (__ballot(spill)) {                       // rare overflow path, wave-uniform branch
      int4 rec = make_int4(0, 0, 0, 0);
      const bool first = spill && mark_block(m, key, f.frame_id, &rec);
      view_append(cnt, view_list, list_cap, first, rec, lane);
    }
    int a = 0;
    if (tmax[1] < tmax[a]) a = 1;
    if (tmax[2] < tmax[a]) a = 2;
    // (select without dynamic register indexing)
    if (a == 0) { cur[0] += step[0]; tmax[0] = tmax[0] + tdelta[0]; }
    else if (a == 1) { cur[1] += step[1]; tmax[1] = tmax[1] + tdelta[1]; }
    else { cur[2] += step[2]; tmax[2] = tmax[2] + tdelta[2]; }
  }
  __syncthreads();
  // ---- phase 2: compact the set (ballot + popcount), then one key per lane goes to HBM: the hash probe, the stamp
  //      exchange and the slot read of all the tile's blocks are in flight together instead of one per ray step.
  int32_t nk = 0;
#pragma unroll
  for (int i = 0; i < LSET / 64; i++) {
    const u64 key = lset[i * 64 + lane];
    const u64 mask = __ballot(key != KEY_EMPTY);
    if (key != KEY_EMPTY) lkeys[nk + (int32_t)__popcll(mask & ((1ull << lane) - 1ull))] = key;
    nk += (int32_t)__popcll(mask);
  }
  __syncthreads();
  for (int32_t i = 0; i < nk; i += 64) {
    int4 rec = make_int4(0, 0, 0, 0);
    const bool first = (i + lane < nk) && mark_block(m, lkeys[i + lane], f.frame_id, &rec);
    view_append(cnt, view_list, list_cap, first, rec, lane);
  }
}

// Dependent-access chain: {view count, view record} -> {depth gather, voxel} -> store.  The record of the first block
// is fetched speculatively beside the count, the voxel is fetched before the projection decides whether it is needed,
// and the flag / dirty-list atomics of lane 0 are issued first and consumed last.
template <typename Img>
__global__ __launch_bounds__(512) void k_integrate_tsdf(DMap m, Frame f, Img depth, const int4* view_list, int32_t list_cap,
                                                        int32_t* esdf_dirty, int32_t* mesh_dirty, int32_t mesh_cnt) {
  int4 rec = view_list[blockIdx.x];                       // speculative: valid iff blockIdx.x < n (gridDim.x <= list_cap)
  int32_t n = m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  if (n > list_cap) n = list_cap;
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
    if (i != (int32_t)blockIdx.x) rec = view_list[i];
    const uint32_t slot = (uint32_t)rec.x;               // pool slot (stable across hash rebuilds)
    if (!slot_ok(slot)) continue;
    float2* vp = &m.tsdf[(size_t)slot * 512 + tid];
    const float2 cur = *vp;
    uint32_t old = 0;
    if (tid == 0) old = atomicOr(&m.slot_flags[slot], F_TSDF | F_DIRTY_ESDF | F_DIRTY_MESH);
    float pc[3];
    apply_rt(f.R_CL, f.t_CL, voxel_center(rec.y, vx, f.block_size, f.voxel_size), voxel_center(rec.z, vy, f.block_size, f.voxel_size),
             voxel_center(rec.w, vz, f.block_size, f.voxel_size), pc);
    float u, v, ds;
    bool upd = cam_project(f, pc, &u, &v);
    const float vd = pc[2];
    if (upd && f.max_dist > 0.0f && vd > f.max_dist) upd = false;
    if (upd) upd = interp_depth(depth, f.rows, f.cols, u, v, f.interp_nearest, &ds);
    if (upd) {
      const float sdf = ds - vd;
      if (!(sdf < -f.trunc)) {
        const float wm = weight_fn(f.weighting_mode, ds, vd, f.trunc);
        const float wsum = wm + cur.y;
        if (wsum > 0.0f) {
          float fused = (sdf * wm + cur.x * cur.y) / wsum;
          if (fused > 0.0f) fused = fminf(f.trunc, fused); else fused = fmaxf(-f.trunc, fused);
          *vp = make_float2(fused, fminf(wsum, f.max_weight));
        }
      }
    }
    if (tid == 0) {
      if (!(old & F_DIRTY_ESDF)) esdf_dirty[atomicAdd(&m.counters[C_ESDF_DIRTY], 1)] = (int32_t)slot;
      if (!(old & F_DIRTY_MESH)) mesh_dirty[atomicAdd(&m.counters[mesh_cnt], 1)] = (int32_t)slot;
    }
  }
}

template <typename Img>
static int integrate_depth_impl(nvbx_mapper* m, Img img, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera) {
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;     // k_integrate_tsdf writes what k_esdf_mark reads
  m->frame_id++;
  Frame f = m->make_frame(T_L_C, camera, rows, cols, m->p.raycast_subsampling_factor);
  const int s = f.subsample;
  f.n_ray_rows = (rows + s - 1 + s - 1) / s;   // indices i with i*s < rows + s - 1
  f.n_ray_cols = (cols + s - 1 + s - 1) / s;
  const int tiles = ((f.n_ray_rows + 7) / 8) * ((f.n_ray_cols + 7) / 8);
  NVBX_LAUNCH(m, (k_mark_view<Img>), dim3(tiles), dim3(64), m->d, f, img, (int4*)m->view_list, (int32_t)m->capacity);
  const int grid = (int)std::min<int64_t>(m->capacity, 1024);
  NVBX_LAUNCH(m, (k_integrate_tsdf<Img>), dim3(grid), dim3(512), m->d, f, img, (const int4*)m->view_list, (int32_t)m->capacity,
                     m->esdf_dirty, m->mesh_dirty_live(), m->mesh_dirty_counter());
  NVBX_HIP(hipGetLastError());
  m->last_view_frame = m->frame_id;
  return m->mark_main();
}

extern "C" int nvbx_integrate_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                    const nvbx_camera* camera) {
  if (!m || !depth_dev || !T_L_C || !camera || rows <= 0 || cols <= 0) { set_error("nvbx_integrate_depth: invalid argument"); return NVBX_E_INVALID; }
  return integrate_depth_impl(m, DepthF32{depth_dev}, rows, cols, T_L_C, camera);
}
extern "C" int nvbx_integrate_depth_u16mm(nvbx_mapper* m, const uint16_t* depth_mm_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                          const nvbx_camera* camera) {
  if (!m || !depth_mm_dev || !T_L_C || !camera || rows <= 0 || cols <= 0) { set_error("nvbx_integrate_depth_u16mm: invalid argument"); return NVBX_E_INVALID; }
  return integrate_depth_impl(m, DepthU16mm{depth_mm_dev}, rows, cols, T_L_C, camera);
}
