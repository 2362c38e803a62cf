// maintenance.hip -- Mapper::decayTsdf / clearOutsideRadius on MI355X (device-side deallocation + hash rebuild).
//
// Deallocation never leaves tombstones: freed slots go back on the free stack (their 4 KiB blocks are zeroed by the
// freeing workgroup, so a popped slot is always clean) and the hash table is rebuilt on the device from the live slots --
// radius clearing: stamps saved, memset, one insert per live slot, the ESDF layer's AABB recomputed (three launches);
// decay (round 6): k_decay itself enters every slot it leaves live into an all-empty table that becomes the live one behind
// the launch, and empties the table of the call before for the call after (three tables rotate: no launch follows a decay) (tombstones instead were measured: the workload frees and re-allocates the same blocks, the
// probe chains grow -- EXPERIMENTS.md).  Call sites served: nvblox_ros/src/lib/nvblox_node.cpp:931-936 (decayTsdf...),
// :1566-1583 (clearOutsideRadius); parameters nvblox_base.yaml:103-107.
#include <algorithm>
#include <cstring>
#include <vector>
#include "nvbx_mapper.h"

using namespace nvbx;

constexpr uint32_t LAYER_MASK = F_TSDF | F_COLOR | F_ESDF | F_MESH | F_ESDF_PENDING | F_FREESPACE;   // a slot carrying any of these is live

__device__ inline void free_slot(DMap& m, uint32_t slot) {   // one thread
  const int32_t pos = atomicAdd(&m.counters[C_FREE_TOP], 1);
  m.free_stack[pos] = slot;
  m.slot_consumed[slot] = STAMP_NEVER; m.slot_stamp[slot] = STAMP_NEVER; m.slot_cam[slot] = STAMP_NEVER;
}

// A block whose weights all fall below the threshold is deallocated.  If it lies in the ESDF z band and its column already has
// an ESDF block, that column is flagged for a re-mark (F_ESDF_REMARK) and put on the ESDF work list -- the flag lives on the
// ESDF slot, which survives, not on the TSDF slot, which may be freed and recycled before the update runs.
//
// HBM-streaming form, no barriers and no LDS: ONE WAVEFRONT PER BLOCK.  A lane holds eight voxels of its block (register r = the
// x = r slab: eight 512-byte loads per wavefront in flight), so "does any voxel survive" is a ballot and the eight band bits of the
// block come out of the same registers.  The per-block bookkeeping (dirty flags, work lists, deallocation) is taken 64 blocks at a
// time: lane j fetches the flags (and the exclusion stamp) of the wavefront's j-th next block before the voxel loop and keeps that
// block's books after it -- the returning flag atomics of 64 blocks are one round trip, and every work list gets ONE
// wave-aggregated reservation per 64 blocks.  (Steady state on the 146 k-block map: 235 us = 5.1 TB/s; the workgroup-synchronised form
// before it 249 us; a bare read-modify-write of the same 1.2 GB in this access pattern 206 us -- tools/stream_ceiling.hip)
__device__ inline void wave_list_append(const DMap& m, int32_t list, bool push, int32_t v, int lane) {      // whole wavefront
  const u64 mask = __ballot(push);
  if (!mask) return;
  const int sh = my_shard();
  int32_t base = 0;
  const int leader = __ffsll((long long)mask) - 1;
  if (lane == leader) base = atomicAdd(shc_at(m, list, sh, 0), (int32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (push) {
    const int32_t p = base + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (p < (int32_t)m.capacity) m.lists[((size_t)list * NSH + sh) * m.capacity + p] = v;
  }
}
// OCC = false: TSDF decay (weight *= factor; a block lives while any weight >= thresh).  OCC = true: occupancy decay (`factor` /
// `thresh` carry the log-odds steps of the free / occupied regions; every value moves towards 0 and stops there; a block lives
// while any value != 0) -- [U] OccupancyDecayIntegrator, Mapper::decayOccupancyAllVoxels (nvblox_node.cpp:925-929).
template <bool OCC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_decay(DMap m, float factor, float thresh, uint32_t exclude_stamp, uint32_t exclude_mask, int32_t mesh_list,
                                               int32_t bz_lo, int32_t bz_hi, int32_t bz_out, float trunc, int32_t* cleared_idx, int32_t keep_blocks, int32_t to_free, float free_dist, Entry* next_table,
                                               Entry* clear_table, int32_t n_clear_wg) {
  // [U] decay switches (mapper_initialization.cpp:383-428; restated line by line in the CPU checker): keep_blocks =
  // !decay_integrator_deallocate_decayed_blocks (a fully decayed block stays allocated); to_free = tsdf_set_free_distance_on_decayed (an
  // OBSERVED voxel whose weight falls below the threshold becomes free: distance free_dist, weight = the threshold) resp.
  // occupancy_decay_to_free (occupied voxels decay past unknown into free and stay there; free voxels are not decayed)
  const int tid = threadIdx.x, lane = tid & 63;
  // the last n_clear_wg workgroups empty the table that was live BEFORE the previous decay call (three tables rotate: live, next, and the one being emptied
  // here for the call after this one): no memset launch behind a decay call
  const int32_t n_decay_wg = (int32_t)gridDim.x - n_clear_wg;
  if ((int32_t)blockIdx.x >= n_decay_wg) {
    uint4* t = reinterpret_cast<uint4*>(clear_table);
    const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (uint32_t i = ((uint32_t)blockIdx.x - (uint32_t)n_decay_wg) * 512u + (uint32_t)tid; i <= m.mask; i += (uint32_t)n_clear_wg * 512u) t[i] = ff;
    return;
  }
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int32_t stride = n_decay_wg * 8;                               // wavefronts of the decay part: wavefront g takes slots g, g + stride, ...
  for (int32_t s0 = (int32_t)blockIdx.x * 8 + (tid >> 6); s0 < hw; s0 += 64 * stride) {
    // ---- lane j: flags (and exclusion) of this round's j-th block
    const int32_t mine = s0 + lane * stride;
    const uint32_t fl = mine < hw ? m.slot_flags[mine] : 0u;
    bool act = (fl & F_TSDF) != 0;
    if (exclude_stamp && act) {
      const uint32_t st = m.slot_cam[mine];       // (the camera's own stamp, not Entry::stamp: a LiDAR scan since then has re-claimed that)
      if (stamp_frame(st) == exclude_stamp && (st & exclude_mask)) act = false;
    }
    const u64 act_mask = __ballot(act);
    bool my_alive = false; uint32_t my_band = 0u;
    // ---- the voxels, one block per iteration
#pragma unroll 1
    for (int j = 0; j < 64; j++) {
      if (!((act_mask >> j) & 1ull)) continue;                         // (uniform)
      const int32_t slot = s0 + j * stride;
      const uint32_t bflags = (uint32_t)__builtin_amdgcn_readlane((int)fl, j);
      float2* vp = &m.tsdf[(size_t)slot * 512 + lane];
      float2 tv[8];
#pragma unroll
      for (int r = 0; r < 8; r++) tv[r] = vp[r * 64];
      bool live = false;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (OCC) {
          float v = tv[r].x;
          if (v > 0.0f) { v = v + thresh; if (v < 0.0f && !to_free) v = 0.0f; }
          else if (v < 0.0f && !to_free) { v = v + factor; if (v > 0.0f) v = 0.0f; }
          tv[r] = make_float2(v, 0.0f); live = live || v != 0.0f;
        } else {
          const float w0 = tv[r].y;
          float w = w0 * factor;
          if (!(w < thresh)) live = true;
          else if (to_free && w0 > 0.0f) { tv[r].x = free_dist; w = thresh; }
          tv[r].y = w;
        }
      }
      const bool alive = keep_blocks || __ballot(live) != 0ull;        // (uniform)
      uint32_t band = 0u;
      if (alive) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
          vp[r * 64] = tv[r];
          if (!OCC && __ballot(in_band(tv[r].x, tv[r].y, trunc)) != 0ull) band |= 1u << (F_BAND_SHIFT + r);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++) vp[r * 64] = make_float2(0.0f, 0.0f);
        if (!OCC) {                                        // (occupancy mappers carry neither colour nor freespace)
          uint2* cp = &m.color[(size_t)slot * 512 + lane];
#pragma unroll
          for (int r = 0; r < 8; r++) cp[r * 64] = make_uint2(0u, 0u);
          if (bflags & F_FREESPACE) {
            int4* fp = &m.freespace[(size_t)slot * 512 + lane];
#pragma unroll
            for (int r = 0; r < 8; r++) fp[r * 64] = make_int4(0, 0, 0, 0);
          }
        }
      }
      if (lane == j) { my_alive = alive; my_band = band; }
    }
    // ---- the books, lane j for block j
    const int32_t slot = mine;
    bool push_esdf = false, push_esdf2 = false, push_mesh = false, push_cleared = false; int32_t esdf2 = 0;
    if (act) {
      uint32_t old = F_DIRTY_MESH;                       // (occupancy: no mesh, nothing joins the mesh list)
      if (my_alive) {
        if (!OCC) { atomicAnd(&m.slot_flags[slot], ~(F_BAND | F_BAND_STALE)); }      // all eight band bits are rewritten: exact again
        const uint32_t o = atomicOr(&m.slot_flags[slot], OCC ? F_DIRTY_ESDF : (F_DIRTY_ESDF | F_DIRTY_MESH | my_band));
        if (!OCC) old = o;
        push_esdf = !(o & F_DIRTY_ESDF);
      } else {
        if (!OCC) old = atomicOr(&m.slot_flags[slot], F_DIRTY_MESH);
        push_cleared = true;                             // Mapper::getClearedBlocks (layer_publishing.cpp:716): the viewer deletes it
        atomicAnd(&m.slot_flags[slot], ~(F_TSDF | F_COLOR | F_MESH | F_FREESPACE | F_BAND | F_BAND_STALE));
        const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
        if (bz >= bz_lo && bz <= bz_hi) {
          const uint32_t es = bz_out == INT32_MIN ? (uint32_t)slot : any_slot(m, bx, by, bz_out);   // 3-D ESDF: the block's own slot (the table is rebuilt after this kernel, not during it)
          if (slot_ok(es) && (m.slot_flags[es] & F_ESDF)) {
            const uint32_t eold = atomicOr(&m.slot_flags[es], F_ESDF_REMARK | F_DIRTY_ESDF);
            if (!(eold & F_DIRTY_ESDF)) { push_esdf2 = true; esdf2 = (int32_t)es; }
          }
        }
        if (!(fl & (F_ESDF | F_ESDF_PENDING))) { atomicAnd(&m.slot_flags[slot], OCC ? ~(F_DIRTY_ESDF | F_DIRTY_MESH) : ~F_DIRTY_ESDF); free_slot(m, (uint32_t)slot); }
      }
      push_mesh = !(old & F_DIRTY_MESH);
    }
    wave_list_append(m, S_LIST_ESDF_DIRTY, push_esdf, slot, lane);
    wave_list_append(m, S_LIST_ESDF_DIRTY, push_esdf2, esdf2, lane);
    wave_list_append(m, mesh_list, push_mesh, slot, lane);
    {
      const u64 cm = __ballot(push_cleared);
      if (cm) {
        int32_t base = 0;
        const int leader = __ffsll((long long)cm) - 1;
        if (lane == leader) base = atomicAdd(&m.counters[C_CLEARED], (int32_t)__popcll(cm));
        base = __shfl(base, leader);
        if (push_cleared) {
          const int32_t p = base + (int32_t)__popcll(cm & ((1ull << lane) - 1ull));
          if (p < (int32_t)m.capacity) { cleared_idx[3 * p] = m.slot_index[3 * slot]; cleared_idx[3 * p + 1] = m.slot_index[3 * slot + 1]; cleared_idx[3 * p + 2] = m.slot_index[3 * slot + 2]; }
        }
      }
    }
    // ---- the table of the map as this launch leaves it (round 6): the lane that keeps a slot's books also enters the slot -- if it is still live -- into the
    // SPARE table (all-empty, same size), with the view stamp of its old entry; the host swaps the two tables behind the launch.  The rebuild that
    // used to follow every decay call (k_save_stamps, a 2 MB memset, k_reinsert: three launches, ~19 us for a room-sized map) is this -- and the riders at the top of
    // the kernel, which empty the table for the call after this one.  Lookups of this launch (any_slot above) read the OLD table, which nobody writes here.  What is live afterwards: a slot this
    // lane has just freed of its projective layers stays only if it carries the ESDF layer; every other slot as its flags said (no other lane changes layer bits).
    if (next_table && mine < hw) {
      const uint32_t layers_after = (act && !my_alive) ? (fl & (F_ESDF | F_ESDF_PENDING)) : (fl & LAYER_MASK);
      if (layers_after) {
        const int32_t x = m.slot_index[3 * slot], y = m.slot_index[3 * slot + 1], z = m.slot_index[3 * slot + 2];
        const uint32_t stamp = m.table[m.slot_entry[slot]].stamp;
        const u64 key = pack_key(x, y, z);
        uint32_t h = table_pos(m, x, y, z);
        for (;;) {
          const u64 k = atomicCAS(&next_table[h].key, KEY_EMPTY, key);
          if (k == KEY_EMPTY) break;
          h = (h + 1) & m.mask;
        }
        next_table[h].slot = (uint32_t)slot; next_table[h].stamp = stamp;
        m.slot_entry[slot] = h;
      }
    }
  }
}

// `srec`: window record of the next ESDF update.  An ESDF block that is dropped takes its sites with it: the distances of
// every voxel within the search radius of those sites are stale, so the block joins the next update's window (the
// distance transform is exact on any window that contains every change of the site set).
__global__ __launch_bounds__(512) void k_clear_outside(DMap m, float cx, float cy, float cz, float r2, float bs, int32_t srec, int32_t esdf3d, int32_t* cleared_idx) {
  // Two steps per 512 slots: every thread tests ONE slot (coalesced flag / index loads: on a large map almost every block stays, and a
  // workgroup per slot spent two dependent round trips on each -- 92 us for 146 k blocks), the few slots outside are collected in LDS,
  // then the workgroup clears those one after the other with all its threads.
  __shared__ int32_t s_out[512];
  __shared__ uint32_t s_oflags[512];       // the flags as tested: thread 0 clears a slot's flags while other wavefronts may still be on that slot
  __shared__ int32_t s_nout;
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int tid = threadIdx.x;
  for (int32_t base = (int32_t)blockIdx.x * 512; base < hw; base += (int32_t)gridDim.x * 512) {
    __syncthreads();
    if (tid == 0) s_nout = 0;
    __syncthreads();
    {
      const int32_t slot = base + tid;
      const uint32_t flags = slot < hw ? m.slot_flags[slot] : 0u;
      if (flags & LAYER_MASK) {
        const float dx = ((float)m.slot_index[3 * slot] * bs + bs * 0.5f) - cx;
        const float dy = ((float)m.slot_index[3 * slot + 1] * bs + bs * 0.5f) - cy;
        const float dz = ((float)m.slot_index[3 * slot + 2] * bs + bs * 0.5f) - cz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 > r2) { const int32_t q = atomicAdd(&s_nout, 1); s_out[q] = slot; s_oflags[q] = flags; }
      }
    }
    __syncthreads();
    const int32_t nout = s_nout;
    for (int32_t k = 0; k < nout; k++) {
    const int32_t slot = s_out[k];
    const uint32_t flags = s_oflags[k];
    if (flags & F_TSDF) m.tsdf[(size_t)slot * 512 + tid] = make_float2(0.0f, 0.0f);
    if (flags & F_COLOR) m.color[(size_t)slot * 512 + tid] = make_uint2(0u, 0u);
    if (flags & F_ESDF) m.esdf[(size_t)slot * 512 + tid] = make_uint2(0u, 0u);
    if (flags & F_FREESPACE) m.freespace[(size_t)slot * 512 + tid] = make_int4(0, 0, 0, 0);
    if (tid == 0) {
      if (flags & F_TSDF) {                              // Mapper::getClearedBlocks (layer_publishing.cpp:716)
        const int32_t p = atomicAdd(&m.counters[C_CLEARED], 1);
        if (p < (int32_t)m.capacity) { cleared_idx[3 * p] = m.slot_index[3 * slot]; cleared_idx[3 * p + 1] = m.slot_index[3 * slot + 1]; cleared_idx[3 * p + 2] = m.slot_index[3 * slot + 2]; }
      }
      if ((flags & F_ESDF) && esdf3d) {                  // 3-D ESDF: the dropped block joins the next update's 3-D window
        const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
        atomicMin(&m.counters[C_ESDF3_WIN + 0], bx); atomicMin(&m.counters[C_ESDF3_WIN + 1], by); atomicMin(&m.counters[C_ESDF3_WIN + 2], bz);
        atomicMax(&m.counters[C_ESDF3_WIN + 3], bx); atomicMax(&m.counters[C_ESDF3_WIN + 4], by); atomicMax(&m.counters[C_ESDF3_WIN + 5], bz);
      } else if ((flags & F_ESDF) && m.site_bits[slot] != 0ull) {
        const int sh = my_shard(); const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1];
        atomicMin(shc_at(m, srec, sh, 0), bx); atomicMin(shc_at(m, srec, sh, 1), by);
        atomicMax(shc_at(m, srec, sh, 2), bx); atomicMax(shc_at(m, srec, sh, 3), by);
      }
      atomicAnd(&m.slot_flags[slot], ~(LAYER_MASK | F_DIRTY_ESDF | F_DIRTY_MESH | F_ESDF_REMARK | F_BAND | F_BAND_STALE));
      m.site_bits[slot] = 0ull; m.obs_bits[slot] = 0ull; m.inside_bits[slot] = 0ull; free_slot(m, (uint32_t)slot);
    }
    }
  }
}

// Mapper::clearTsdfInsideShapes (nvblox_node.cpp:1834): one workgroup per TSDF block, lane = voxel; the shape list sits in
// kernel-argument space (<= 16 shapes per launch)
struct ShapeArgs { int32_t n; nvbx_bounding_shape s[16]; };
__global__ __launch_bounds__(512) void k_clear_shapes(DMap m, ShapeArgs sh, float vs, float bs, int32_t mesh_list, float trunc, int32_t occupancy) {
  __shared__ int s_touched;
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  for (int32_t slot = blockIdx.x; slot < hw; slot += gridDim.x) {
    if (!(m.slot_flags[slot] & F_TSDF)) continue;
    __syncthreads();
    if (tid == 0) s_touched = 0;
    __syncthreads();
    const float px = voxel_center(m.slot_index[3 * slot], vx, bs, vs), py = voxel_center(m.slot_index[3 * slot + 1], vy, bs, vs),
                pz = voxel_center(m.slot_index[3 * slot + 2], vz, bs, vs);
    bool inside = false;
    for (int k = 0; k < sh.n && !inside; k++) {
      const nvbx_bounding_shape& q = sh.s[k];
      if (q.kind == 0) { const float dx = px - q.a[0], dy = py - q.a[1], dz = pz - q.a[2]; inside = ((dx * dx + dy * dy) + dz * dz) <= q.b[0] * q.b[0]; }
      else inside = px >= q.a[0] && py >= q.a[1] && pz >= q.a[2] && px <= q.b[0] && py <= q.b[1] && pz <= q.b[2];
    }
    float2 fin = m.tsdf[(size_t)slot * 512 + tid];
    if (inside) { fin = make_float2(0.0f, 0.0f); m.tsdf[(size_t)slot * 512 + tid] = fin; s_touched = 1; }
    __syncthreads();
    if (s_touched && !occupancy) publish_band(m.slot_flags, (uint32_t)slot, tid, in_band(fin.x, fin.y, trunc));      // (s_touched is uniform after the barrier)
    if (tid == 0 && s_touched) {
      const uint32_t old = atomicOr(&m.slot_flags[slot], F_DIRTY_ESDF | F_DIRTY_MESH);
      if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, slot);
      if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, slot);
    }
  }
}
extern "C" int nvbx_clear_tsdf_inside_shapes(nvbx_mapper* m, const nvbx_bounding_shape* shapes_host, int32_t n_shapes) {
  if (!m || n_shapes < 0 || (n_shapes > 0 && !shapes_host)) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  const int grid = (int)std::min<int64_t>(m->capacity, 1024);
  for (int32_t o = 0; o < n_shapes; o += 16) {
    ShapeArgs a{}; a.n = std::min(16, n_shapes - o);
    for (int i = 0; i < a.n; i++) a.s[i] = shapes_host[o + i];
    NVBX_LAUNCH(m, k_clear_shapes, dim3(grid), dim3(512), m->d, a, m->p.voxel_size, m->p.voxel_size * 8.0f, m->mesh_list_live(),
                m->p.truncation_distance_vox * m->p.voxel_size, (int32_t)(m->p.projective_layer_type == 1));
  }
  NVBX_HIP(hipGetLastError());
  return m->mark_main();
}

// rebuild: (1) remember each live slot's view stamp, (2) memset table, (3) re-insert live slots, recompute ESDF AABB
__global__ void k_save_stamps(DMap m, uint32_t* tmp) {
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < hw; s += gridDim.x * blockDim.x)
    tmp[s] = (m.slot_flags[s] & LAYER_MASK) ? m.table[m.slot_entry[s]].stamp : 0xFFFFFFFFu;
  if (blockIdx.x == 0 && threadIdx.x < 4) m.counters[C_ESDF_AABB + threadIdx.x] = threadIdx.x < 2 ? INT32_MAX : INT32_MIN;
}
__global__ void k_reinsert(DMap m, const uint32_t* tmp, int32_t bz_out) {
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < hw; s += gridDim.x * blockDim.x) {
    const uint32_t flags = m.slot_flags[s];
    if (!(flags & LAYER_MASK)) continue;
    const int32_t x = m.slot_index[3 * s], y = m.slot_index[3 * s + 1], z = m.slot_index[3 * s + 2];
    const u64 key = pack_key(x, y, z);
    uint32_t h = table_pos(m, x, y, z);
    for (;;) {
      const u64 k = atomicCAS(&m.table[h].key, KEY_EMPTY, key);
      if (k == KEY_EMPTY) break;
      h = (h + 1) & m.mask;
    }
    m.table[h].slot = (uint32_t)s; m.table[h].stamp = tmp[s];
    m.slot_entry[s] = h;
    if ((flags & F_ESDF) && z == bz_out) {       // the slicer's image covers the ESDF blocks of the slice plane
      // (an atomic only where the block extends the box as last seen: thousands of ESDF blocks on four addresses serialise otherwise;
      // a stale read costs an unnecessary atomic, never a missing one)
      int32_t* bb = &m.counters[C_ESDF_AABB];
      if (x < __hip_atomic_load(&bb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&bb[0], x);
      if (y < __hip_atomic_load(&bb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&bb[1], y);
      if (x > __hip_atomic_load(&bb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bb[2], x);
      if (y > __hip_atomic_load(&bb[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&bb[3], y);
    }
  }
}

#ifndef NVBX_DECAY_INLINE_REBUILD
#define NVBX_DECAY_INLINE_REBUILD 1       // (A/B: 0 = the three-launch rebuild behind every decay call, as up to round 5)
#endif
// Three hash tables of one size rotate through a mapper that decays: LIVE (DMap::table), NEXT (all-empty: k_decay enters the surviving slots) and DIRTY (the table
// that was live before the previous call: emptied by riders of this call's k_decay, the next call's NEXT).  Allocated at the first two decay calls, again after a
// growth.  *clear_out = nullptr: nothing to empty in this launch (first call).
static int prepare_tables(nvbx_mapper* m, Entry** next_out, Entry** clear_out) {
  const size_t bytes = ((size_t)m->d.mask + 1) * sizeof(Entry);
  if (m->table_mask_extra != m->d.mask) {               // first decay, or the table has grown
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->table_spare) NVBX_HIP(hipFree(m->table_spare));
    if (m->table_dirty) NVBX_HIP(hipFree(m->table_dirty));
    m->table_spare = nullptr; m->table_dirty = nullptr; m->table_mask_extra = m->d.mask;
  }
  if (!m->table_spare) {
    NVBX_HIP(hipMalloc(&m->table_spare, bytes));
    NVBX_HIP(hipMemsetAsync(m->table_spare, 0xFF, bytes, m->stream));
  }
  *next_out = static_cast<Entry*>(m->table_spare); *clear_out = static_cast<Entry*>(m->table_dirty);
  return NVBX_OK;
}
// behind the k_decay launch: NEXT becomes live, the table emptied by the launch's riders becomes the next call's NEXT, the old live table waits to be emptied
static int rotate_tables(nvbx_mapper* m) {
  Entry* old_live = m->d.table;
  m->d.table = static_cast<Entry*>(m->table_spare);
  m->table_spare = m->table_dirty;          // (emptied by this launch; nullptr after the first call: prepare_tables allocates a second one)
  m->table_dirty = old_live;
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
static int rebuild_table(nvbx_mapper* m) {
  uint32_t* tmp = (uint32_t*)m->export_idx;   // capacity * 12 bytes scratch >= capacity * 4
  const unsigned rg = (unsigned)std::min<int64_t>((m->capacity + 255) / 256, 2048);       // one thread per slot up to 512 k slots
  NVBX_LAUNCH(m, k_save_stamps, dim3(rg), dim3(256), m->d, tmp);
  NVBX_HIP(hipMemsetAsync(m->d.table, 0xFF, ((size_t)m->d.mask + 1) * sizeof(Entry), m->stream));
  NVBX_LAUNCH(m, k_reinsert, dim3(rg), dim3(256), m->d, tmp, m->make_esdf_args().bz_out);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

extern "C" int nvbx_decay_occupancy(nvbx_mapper* m) {
  if (!m) return NVBX_E_INVALID;
  if (m->p.projective_layer_type != 1) { set_error("nvbx_decay_occupancy: not an occupancy mapper"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->undo_marks()) return NVBX_E_DEVICE;
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  EsdfArgs ea = m->make_esdf_args();
  if (m->p.esdf_mode == 1) { ea.bz_lo = INT32_MIN + 1; ea.bz_hi = INT32_MAX; ea.bz_out = INT32_MIN; }      // 3-D ESDF: every block is its own column
  else if (ea.plane_on) { ea.bz_lo = INT32_MIN + 1; ea.bz_hi = INT32_MAX; }      // ground-plane mode: the band's blocks vary with (x, y) -- a deallocated block of ANY height has its column re-marked
  const int64_t hw_seen = std::max<int64_t>(1, __atomic_load_n(&m->h_mirror[1], __ATOMIC_RELAXED));       // (a hint: the kernel grid-strides)
  const bool inline_rebuild = NVBX_DECAY_INLINE_REBUILD != 0;
  Entry* next_table = nullptr; Entry* clear_table = nullptr;
  if (inline_rebuild && prepare_tables(m, &next_table, &clear_table)) return NVBX_E_DEVICE;
  const int32_t n_clear_wg = clear_table ? 64 : 0;
  NVBX_LAUNCH(m, k_decay<true>, dim3((unsigned)std::min<int64_t>(std::min<int64_t>(m->capacity, 4096), std::max<int64_t>(512, (hw_seen + 7) / 8)) + (unsigned)n_clear_wg), dim3(512), m->d,
              log_odds(m->p.free_region_decay_probability), log_odds(m->p.occupied_region_decay_probability), 0u, 0u, (int32_t)m->mesh_list_live(),
              ea.bz_lo, ea.bz_hi, ea.bz_out, 0.0f, m->cleared_idx, (int32_t)(m->p.decay_deallocate_decayed_blocks ? 0 : 1), (int32_t)(m->p.occupancy_decay_to_free ? 1 : 0), 0.0f,
              next_table, clear_table, n_clear_wg);
  return inline_rebuild ? rotate_tables(m) : rebuild_table(m);
}

extern "C" int nvbx_decay_tsdf(nvbx_mapper* m, int32_t exclude_last_view) {
  if (!m) return NVBX_E_INVALID;
  if (m->p.projective_layer_type == 1) { set_error("nvbx_decay_tsdf: occupancy mapper (use nvbx_decay_occupancy)"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->undo_marks()) return NVBX_E_DEVICE;          // decay deallocates: unresolved marking passes are taken back first
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  static const int decay_grid = getenv("NVBX_DECAY_GRID") ? atoi(getenv("NVBX_DECAY_GRID")) : 4096;      // (env: tools/decay_grid_sweep.sh)
  // eight slots per workgroup iteration: a room-sized map needs a few hundred workgroups, and launching 4096 costs it ~1.5 us
  const int64_t hw_seen = std::max<int64_t>(1, __atomic_load_n(&m->h_mirror[1], __ATOMIC_RELAXED));       // (high-water mark as last reported: a hint, the kernel grid-strides)
  const int grid = (int)std::min<int64_t>(std::min<int64_t>(m->capacity, decay_grid), std::max<int64_t>(512, (hw_seen + 7) / 8));
  EsdfArgs ea = m->make_esdf_args();
  if (m->p.esdf_mode == 1) { ea.bz_lo = INT32_MIN + 1; ea.bz_hi = INT32_MAX; ea.bz_out = INT32_MIN; }      // 3-D ESDF: every block is its own column
  else if (ea.plane_on) { ea.bz_lo = INT32_MIN + 1; ea.bz_hi = INT32_MAX; }      // ground-plane mode: the band's blocks vary with (x, y) -- a deallocated block of ANY height has its column re-marked
  const bool inline_rebuild = NVBX_DECAY_INLINE_REBUILD != 0;
  Entry* next_table = nullptr; Entry* clear_table = nullptr;
  if (inline_rebuild && prepare_tables(m, &next_table, &clear_table)) return NVBX_E_DEVICE;
  const int32_t n_clear_wg = clear_table ? 64 : 0;
  NVBX_LAUNCH(m, k_decay<false>, dim3((unsigned)(grid + n_clear_wg)), dim3(512), m->d, m->p.tsdf_decay_factor, m->p.tsdf_decayed_weight_threshold,
                     exclude_last_view ? m->last_camera_view_frame : 0u, m->last_camera_view_mask, m->mesh_list_live(), ea.bz_lo, ea.bz_hi, ea.bz_out,
                     m->p.truncation_distance_vox * m->p.voxel_size, m->cleared_idx, (int32_t)(m->p.decay_deallocate_decayed_blocks ? 0 : 1),
                     (int32_t)(m->p.tsdf_set_free_distance_on_decayed ? 1 : 0), m->p.tsdf_decayed_free_distance_vox * m->p.voxel_size,
                     next_table, clear_table, n_clear_wg);
  return inline_rebuild ? rotate_tables(m) : rebuild_table(m);
}

extern "C" int nvbx_clear_outside_radius(nvbx_mapper* m, const float center[3], float radius) {
  if (!m || !center) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->undo_marks()) return NVBX_E_DEVICE;          // deallocates: unresolved marking passes are taken back first
  const int grid = (int)std::min<int64_t>((m->capacity + 511) / 512, 2048);          // 512 slots per workgroup iteration
  NVBX_LAUNCH(m, k_clear_outside, dim3(grid), dim3(512), m->d, center[0], center[1], center[2], radius * radius, m->p.voxel_size * 8.0f,
              (int32_t)(S_ESDF_REC + (int)(m->esdf_epoch & 1)), (int32_t)(m->p.esdf_mode == 1), m->cleared_idx);
  return rebuild_table(m);
}

// Mapper::getClearedBlocks (layer_publishing.cpp:716): projective-layer blocks deallocated (decay, radius clearing) since the last call
__global__ void k_zero_cleared(DMap m) { m.counters[C_CLEARED] = 0; }
extern "C" int64_t nvbx_take_cleared_blocks(nvbx_mapper* m, nvbx_index3d* out, int64_t capacity) {
  if (!m || capacity < 0 || (capacity > 0 && !out)) return NVBX_E_INVALID;
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  int64_t n = m->h_counters[C_CLEARED];
  if (n > m->capacity) n = m->capacity;       // (more deallocations than the pool has blocks between two calls: the oldest ones win)
  if (n == 0) return 0;
  std::vector<nvbx_index3d> tmp((size_t)n);
  NVBX_HIP(hipMemcpy(tmp.data(), m->cleared_idx, (size_t)n * 12, hipMemcpyDeviceToHost));
  auto less = [](const nvbx_index3d& a, const nvbx_index3d& b) { if (a.x != b.x) return a.x < b.x; if (a.y != b.y) return a.y < b.y; return a.z < b.z; };
  std::sort(tmp.begin(), tmp.end(), less);
  tmp.erase(std::unique(tmp.begin(), tmp.end(), [](const nvbx_index3d& a, const nvbx_index3d& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }), tmp.end());
  const int64_t u = (int64_t)tmp.size();
  if (u > capacity) return u;                  // too small: nothing is consumed, the caller comes back with room for u
  memcpy(out, tmp.data(), (size_t)u * 12);
  NVBX_LAUNCH(m, k_zero_cleared, dim3(1), dim3(1), m->d);
  NVBX_HIP(hipGetLastError());
  return u;
}

// ------------------------------------------------------------------------------------------------ pool growth
// The reference allocates voxel blocks on demand (Layer::allocateBlockAtIndex); here the pools are flat HBM arrays addressed by slot
// id, so "on demand" = the arrays double before they run out.  Every capacity-sized array is re-allocated at the new size, its
// contents copied (slot ids, and with them every list entry and record, stay valid), the new slots are pushed on the free stack and
// the hash table is rebuilt at twice the size on the device.  Rare (a doubling), so it simply synchronises the stream.
__global__ void k_grow_free_stack(DMap m, uint32_t old_cap, uint32_t new_cap) {
  const int32_t top = m.counters[C_FREE_TOP];
  const uint32_t added = new_cap - old_cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < added; i += gridDim.x * blockDim.x) m.free_stack[top + i] = new_cap - 1u - i;   // the smallest new id pops first
}
__global__ void k_grow_commit(DMap m, int32_t added) { m.counters[C_FREE_TOP] += added; m.host_mirror[0] = m.counters[C_FREE_TOP]; }
__global__ void k_remap_mesh_records(MeshRecord* rec, int32_t n, int64_t old_vreg, int64_t new_vreg, int64_t old_treg, int64_t new_treg) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    MeshRecord r = rec[i];
    if (r.x == INT32_MIN || r.vbase < 0) continue;
    const int64_t sv = r.vbase / old_vreg, st = r.tbase / old_treg;
    r.vbase = (int32_t)(sv * new_vreg + (r.vbase - sv * old_vreg)); r.tbase = (int32_t)(st * new_treg + (r.tbase - st * old_treg));
    rec[i] = r;
  }
}

namespace {
struct PoolArr { void** p; size_t bytes_per_block; int fill; };
template <typename T> PoolArr arr(T** p, size_t bpb, int fill) { return PoolArr{reinterpret_cast<void**>(p), bpb, fill}; }
}  // namespace

int nvbx_mapper::grow_map(int64_t new_cap) {
  if (new_cap <= capacity) return NVBX_OK;
  if (new_cap > (1ll << 24)) new_cap = 1ll << 24;
  const int64_t old_cap = capacity;
  if (join_side()) return NVBX_E_DEVICE;                     // held-back launches go first: they were sized for the old arrays
  if (fetch_counters()) return NVBX_E_DEVICE;                // (synchronises; the mesh arena cursors are needed below)
  // enough HBM?  (voxel pools dominate: 3-4 x 4 KiB per block; old and new copies of ONE array coexist at a time)
  { size_t free_b = 0, total_b = 0; NVBX_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)new_cap * 4096 + ((size_t)new_cap - (size_t)old_cap) * (3 * 4096 + (d.freespace ? 8192 : 0) + 512) + (64u << 20);
    if (free_b < need) { set_error("pool growth: not enough free HBM, the block pools stay at their size"); max_capacity = capacity; return NVBX_OK; } }
  // Failure-atomic: EVERY new array is allocated first; if one allocation fails they are all released and the map is untouched
  // (the pools stay at their size).  Only then are the contents copied and the pointers / capacity swapped in one commit step.
  std::vector<PoolArr> arrs = {
      arr(&d.free_stack, 4, -1), arr(&d.slot_flags, 4, 0), arr(&d.slot_index, 12, 0), arr(&d.slot_entry, 4, 0), arr(&d.slot_stamp, 4, 0xFF),
      arr(&d.slot_consumed, 4, 0xFF), arr(&d.slot_cam, 4, 0xFF), arr(&d.tsdf, 4096, 0), arr(&d.color, 4096, 0), arr(&d.esdf, 4096, 0), arr(&view_list, 16, -1),
      arr(&export_idx, 12, -1), arr(&cleared_idx, 12, -1), arr(&d.site_bits, 8, 0), arr(&d.obs_bits, 8, 0), arr(&d.inside_bits, 8, 0),
      arr(&mesh_rec, sizeof(MeshRecord), -1)};
  if (d.freespace) arrs.push_back(arr(&d.freespace, 512 * 16, 0));
  const int64_t new_vcap = std::min<int64_t>(new_cap * 192, 48ll << 20), new_tcap = new_vcap * 2;
  const bool grow_mesh = new_vcap > mesh_vert_cap;
  uint64_t tsz = 1; while (tsz < (uint64_t)new_cap * 2) tsz <<= 1;
  std::vector<void*> fresh;                                    // everything allocated so far (released on failure)
  auto take = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    fresh.push_back(q); return q;
  };
  bool ok = true;
  uint32_t* saved = (uint32_t*)take((size_t)old_cap * sizeof(uint32_t)); ok = ok && saved;
  std::vector<void*> np(arrs.size(), nullptr);
  for (size_t i = 0; ok && i < arrs.size(); i++) { np[i] = take(arrs[i].bytes_per_block * (size_t)new_cap); ok = ok && np[i]; }
  int32_t* nl = ok ? (int32_t*)take((size_t)N_LISTS * NSH * (size_t)new_cap * 4) : nullptr; ok = ok && nl;
  float* nv = nullptr; float* nn = nullptr; uint8_t* nc = nullptr; int32_t* nt = nullptr;
  if (ok && grow_mesh) {
    nv = (float*)take(new_vcap * 12); nn = (float*)take(new_vcap * 12); nc = (uint8_t*)take(new_vcap * 4); nt = (int32_t*)take(new_tcap * 12);
    ok = nv && nn && nc && nt;
  }
  Entry* ntab = ok ? (Entry*)take(tsz * sizeof(Entry)) : nullptr; ok = ok && ntab;
  if (!ok) {
    for (void* q : fresh) (void)hipFree(q);
    set_error("pool growth: device allocation failed, the block pools stay at their size"); max_capacity = capacity;
    return NVBX_OK;
  }
  // from here on only copies, launches and frees: a failure is a device error, not a half-grown map
  const unsigned rg = (unsigned)std::min<int64_t>((old_cap + 255) / 256, 2048);
  NVBX_LAUNCH(this, k_save_stamps, dim3(rg), dim3(256), d, saved);          // view stamps live in the table that is about to be replaced
  for (size_t i = 0; i < arrs.size(); i++) {
    const PoolArr& a = arrs[i];
    NVBX_HIP(hipMemcpyAsync(np[i], *a.p, a.bytes_per_block * (size_t)old_cap, hipMemcpyDeviceToDevice, stream));
    if (a.fill >= 0) NVBX_HIP(hipMemsetAsync((char*)np[i] + a.bytes_per_block * (size_t)old_cap, a.fill, a.bytes_per_block * (size_t)(new_cap - old_cap), stream));
  }
  // work lists: [list][shard][capacity] -- every segment moves to its place in the wider layout (entry counts live in d.shc)
  for (int q = 0; q < N_LISTS * NSH; q++)
    NVBX_HIP(hipMemcpyAsync(nl + (size_t)q * new_cap, d.lists + (size_t)q * old_cap, (size_t)old_cap * 4, hipMemcpyDeviceToDevice, stream));
  if (grow_mesh) {   // mesh arenas: NSH shard regions each; the used prefix of every region (last mesh update) moves, the records are re-based
    const int64_t ovr = mesh_vert_cap / NSH, otr = mesh_tri_cap / NSH, nvr = new_vcap / NSH, ntr = new_tcap / NSH;
    if (mesh_epoch) {
      const int par = (int)((mesh_epoch + 1) & 1);
      for (int sh = 0; sh < NSH; sh++) {
        const int64_t uv = std::min<int64_t>((uint32_t)h_shc[((S_MESH_REC + par) * NSH + sh) * SH_STRIDE + 2], ovr);
        const int64_t ut = std::min<int64_t>((uint32_t)h_shc[((S_MESH_REC + par) * NSH + sh) * SH_STRIDE + 3], otr);
        if (uv) {
          NVBX_HIP(hipMemcpyAsync(nv + sh * nvr * 3, mesh_vert + sh * ovr * 3, (size_t)uv * 12, hipMemcpyDeviceToDevice, stream));
          NVBX_HIP(hipMemcpyAsync(nn + sh * nvr * 3, mesh_nrm + sh * ovr * 3, (size_t)uv * 12, hipMemcpyDeviceToDevice, stream));
          NVBX_HIP(hipMemcpyAsync(nc + sh * nvr * 4, mesh_col + sh * ovr * 4, (size_t)uv * 4, hipMemcpyDeviceToDevice, stream));
        }
        if (ut) NVBX_HIP(hipMemcpyAsync(nt + sh * ntr * 3, mesh_tri + sh * otr * 3, (size_t)ut * 12, hipMemcpyDeviceToDevice, stream));
      }
      const int32_t nraw = h_counters[C_MESH_OUT + 4 * par + 0];
      // (the records are re-based in their NEW array: np[] of mesh_rec has its copy already enqueued on the same stream)
      MeshRecord* new_rec = nullptr;
      for (size_t i = 0; i < arrs.size(); i++) if (arrs[i].p == reinterpret_cast<void**>(&mesh_rec)) new_rec = (MeshRecord*)np[i];
      if (nraw > 0) NVBX_LAUNCH(this, k_remap_mesh_records, dim3(64), dim3(256), new_rec, std::min<int32_t>(nraw, (int32_t)old_cap), ovr, nvr, otr, ntr);
    }
  }
  NVBX_HIP(hipMemsetAsync(ntab, 0xFF, tsz * sizeof(Entry), stream));
  NVBX_HIP(hipStreamSynchronize(stream));
  // ---- commit: pointers, capacity, hash geometry
  for (size_t i = 0; i < arrs.size(); i++) { (void)hipFree(*arrs[i].p); *arrs[i].p = np[i]; }
  (void)hipFree(d.lists); d.lists = nl;
  if (grow_mesh) {
    (void)hipFree(mesh_vert); (void)hipFree(mesh_nrm); (void)hipFree(mesh_col); (void)hipFree(mesh_tri);
    mesh_vert = nv; mesh_nrm = nn; mesh_col = nc; mesh_tri = nt; mesh_vert_cap = new_vcap; mesh_tri_cap = new_tcap;
  }
  (void)hipFree(d.table); d.table = ntab;
  d.mask = (uint32_t)(tsz - 1);
  { uint32_t lg = 0; while ((1ull << lg) < tsz) lg++; d.shift = 32u - lg; }
  d.capacity = (uint32_t)new_cap; capacity = new_cap;
  NVBX_LAUNCH(this, k_grow_free_stack, dim3(256), dim3(256), d, (uint32_t)old_cap, (uint32_t)new_cap);
  NVBX_LAUNCH(this, k_grow_commit, dim3(1), dim3(1), d, (int32_t)(new_cap - old_cap));
  NVBX_LAUNCH(this, k_reinsert, dim3(rg), dim3(256), d, saved, make_esdf_args().bz_out);
  NVBX_HIP(hipStreamSynchronize(stream));
  (void)hipFree(saved);
  if (view_export_cap > 0) view_export_cap = std::min<int64_t>(view_export_cap, capacity);
  growths++;
  return NVBX_OK;
}

// before a frame is enqueued: fewer than half of the slots free (as of the last frame the GPU has finished) -> double
int nvbx_mapper::maybe_grow(int64_t extra_blocks_wanted) {
  if (capacity >= max_capacity) return NVBX_OK;
  const int64_t free_seen = __atomic_load_n(&h_mirror[0], __ATOMIC_RELAXED);
  int64_t target = capacity;
  while (target < max_capacity && (free_seen + (target - capacity) - extra_blocks_wanted) * 2 < target) target *= 2;      // at least half free afterwards
  if (target > max_capacity) target = max_capacity;
  if (target == capacity) return NVBX_OK;
  return grow_map(target);
}
extern "C" int nvbx_mapper_set_max_capacity(nvbx_mapper* m, int64_t max_block_capacity) {
  if (!m) return NVBX_E_INVALID;
  m->max_capacity = std::max<int64_t>(m->capacity, std::min<int64_t>(max_block_capacity, 1ll << 24));
  return NVBX_OK;
}
extern "C" int64_t nvbx_mapper_capacity(nvbx_mapper* m) { return m ? m->capacity : NVBX_E_INVALID; }
