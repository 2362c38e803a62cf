// esdf.hip -- MultiMapper::updateEsdf (EsdfMode::k2D) + EsdfSlicer on MI355X.
//
// The 2-D ESDF slice is an exact Euclidean distance transform with a cut-off radius R = esdf_max_distance_m / voxel
// (the reference's sweep/propagate loop iterates towards the same field; DESIGN.md "ESDF semantics").  The plane of one
// 8x8x8 ESDF block is exactly one wavefront (64 lanes, lane = x + 8y), stored as one contiguous 512-B line, and its
// site mask is exactly one 64-bit word (__ballot) kept per block in `site_bits`.
// Per update, TWO launches, all sizes decided on the device (no host read-back, no scratch image):
//   k_esdf_mark    one wave per dirty TSDF block: insert the ESDF block (x, y, z_slice), de-duplicate columns with an
//                  epoch stamp, scan the TSDF z-band (each lane owns one (x,y) column = 64 contiguous bytes per TSDF
//                  block) -> observed / inside / site flags, site mask = __ballot; grows the dirty window (atomicMin/Max on
//                  the workgroup's shard of the window record).
//   k_esdf_edt     four waves per ESDF block of the window (dirty AABB + R): gathers the site masks of the (2R/8+1)^2
//                  surrounding blocks through the hash into LDS, expands them to a local row bitmap, finds the nearest
//                  site along x by clz/ctz on 64-bit words (row pass, int8 dx in LDS), then minimises dy^2 + dx^2 over dy
//                  in order of increasing |dy| with early exit (column pass, split over the 4 waves and merged by a packed
//                  integer min) and writes {sq, parent, flags} back as one 8-byte store per lane.
// Call sites served: nvblox_ros/src/lib/nvblox_node.cpp:781 (updateEsdf), :836-844 (sliceLayerToDistanceImage),
// :917-919 (occupancyGridFromSliceImage); conversions/esdf_slice_conversions.cu:33-73; esdf_and_gradients_conversions.cu:88-125.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "nvbx_mapper.h"
#include "nvbx_esdf_mark.h"
#include "nvbx_esdf_edt.h"

using namespace nvbx;

__global__ __launch_bounds__(64) void k_esdf_mark(DMap m, EsdfArgs a) { esdf_mark_worker(m, a, (int)blockIdx.x, (int)gridDim.x); esdf_mark_pass_done(m, a, (int)gridDim.x); }

__global__ __launch_bounds__(256) void k_esdf_edt(DMap m, EsdfArgs a) {
  __shared__ EdtShared sh;
  esdf_edt_worker(m, a, (int)blockIdx.x, (int)gridDim.x, &sh);
}

// ------------------------------------------------------------------------------------------------ esdf_propagation = 1
// [U] open choice (nvbx_mapper_params::esdf_propagation): the reference's EsdfIntegrator sweeps parent directions inside blocks and
// across block faces until nothing changes.  That class of algorithm is restated as SYNCHRONOUS 4-neighbour parent propagation to
// its fixed point, restricted to the allocated ESDF blocks of the slice (a voxel can only learn of a site through a chain of
// allocated neighbours): state = packed (sq << 14 | dy + 64 << 7 | dx + 64) of the best known site, integer min = lexicographic
// (sq, dy, dx) min = the tie rule of the exact transform; every round each voxel takes the min over itself and its four axis
// neighbours' sites re-expressed from its own position, subject to the cut-off.  Every update recomputes the whole slice (the
// fixed point of vector propagation depends on history; a full synchronous recompute is the one definition both sides can share).
// Not the benchmark path: three small kernels + a host loop that reads a "changed" flag every PROP_BATCH rounds.
constexpr int PROP_BATCH = 16;
constexpr int32_t PROP_NONE = INT32_MAX;
__device__ inline bool prop_block(const DMap& m, const EsdfArgs& a, int32_t s, uint32_t* flags_out) {
  const uint32_t flags = m.slot_flags[s];
  *flags_out = flags;
  return (flags & (F_ESDF | F_ESDF_PENDING)) && m.slot_index[3 * s + 2] == a.bz_out;
}
__global__ __launch_bounds__(64) void k_esdf_prop_init(DMap m, EsdfArgs a, int32_t plane_a) {
  const int lane = threadIdx.x;
  const int srec = S_ESDF_REC + (int)(a.epoch & 1), srec_next = S_ESDF_REC + (int)((a.epoch + 1) & 1);
  if (blockIdx.x == 0 && lane < NSH) {            // the book-keeping esdf_edt_worker does for the exact transform
    *shc_at(m, srec_next, lane, 0) = INT32_MAX; *shc_at(m, srec_next, lane, 1) = INT32_MAX;
    *shc_at(m, srec_next, lane, 2) = INT32_MIN; *shc_at(m, srec_next, lane, 3) = INT32_MIN;
    *shc_at(m, srec_next, lane, 4) = 0; *shc_at(m, srec_next, lane, 5) = 0;
    *shc_at(m, S_LIST_ESDF_DIRTY, lane, 0) = 0;
    if (lane == 0) m.counters[a.rec_next + 6] = 0;
  }
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x; s < hw; s += gridDim.x) {
    uint32_t flags;
    if (!prop_block(m, a, s, &flags)) continue;      // uniform
    if (lane == 0) {
      if (flags & F_ESDF_REMARK) atomicAnd(&m.slot_flags[s], ~F_ESDF_REMARK);
      if (flags & F_ESDF_PENDING) {
        const int32_t bx = m.slot_index[3 * s], by = m.slot_index[3 * s + 1];
        atomicOr(&m.slot_flags[s], F_ESDF); atomicAnd(&m.slot_flags[s], ~F_ESDF_PENDING);
        atomicMin(&m.counters[C_ESDF_AABB + 0], bx); atomicMin(&m.counters[C_ESDF_AABB + 1], by);
        atomicMax(&m.counters[C_ESDF_AABB + 2], bx); atomicMax(&m.counters[C_ESDF_AABB + 3], by);
      }
      atomicAdd(shc_at(m, srec, my_shard(), 5), 1);
      atomicAdd(&m.counters[a.rec + 6], 64);
    }
    const bool site = (m.site_bits[s] >> lane) & 1ull;
    m.esdf[(size_t)s * 512 + plane_a * 64 + lane] = make_uint2((uint32_t)(site ? ((64 << 7) | 64) : PROP_NONE), 0u);
  }
}
// candidate: the neighbour at offset (ox, oy) knows a site at (dx, dy) from ITSELF -> (dx + ox, dy + oy) from here
__device__ inline int32_t prop_candidate(int32_t nb, int ox, int oy, float max_sq) {
  if (nb == PROP_NONE) return PROP_NONE;
  const int32_t dx = (nb & 127) - 64 + ox, dy = ((nb >> 7) & 127) - 64 + oy;
  const int32_t sq = dx * dx + dy * dy;
  if (!((float)sq <= max_sq) || dx < -63 || dx > 63 || dy < -63 || dy > 63) return PROP_NONE;
  return (sq << 14) | ((dy + 64) << 7) | (dx + 64);
}
__global__ __launch_bounds__(64) void k_esdf_prop_iter(DMap m, EsdfArgs a, int32_t src, int32_t dst, int32_t* changed) {
  const int lane = threadIdx.x;
  const int vx = lane & 7, vy = lane >> 3;
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x; s < hw; s += gridDim.x) {
    uint32_t flags;
    if (!prop_block(m, a, s, &flags)) continue;      // uniform
    const int32_t bx = m.slot_index[3 * s], by = m.slot_index[3 * s + 1];
    // the four face-neighbour blocks (lanes 0..3 probe, every lane gets the slots); only blocks of the ESDF layer take part
    uint32_t ns = SLOT_NONE;
    if (lane < 4) {
      const int ox = lane == 0 ? -1 : (lane == 1 ? 1 : 0), oy = lane == 2 ? -1 : (lane == 3 ? 1 : 0);
      ns = any_slot(m, bx + ox, by + oy, a.bz_out);
      if (slot_ok(ns) && !(m.slot_flags[ns] & F_ESDF)) ns = SLOT_NONE;
    }
    const uint32_t n_xm = __shfl(ns, 0), n_xp = __shfl(ns, 1), n_ym = __shfl(ns, 2), n_yp = __shfl(ns, 3);
    const int32_t own = (int32_t)m.esdf[(size_t)s * 512 + src * 64 + lane].x;
    // neighbours inside the block by shuffle, across a face from the neighbour block's border voxel
    int32_t v_xm = __shfl(own, (lane + 63) & 63), v_xp = __shfl(own, (lane + 1) & 63), v_ym = __shfl(own, (lane + 56) & 63), v_yp = __shfl(own, (lane + 8) & 63);
    if (vx == 0) v_xm = slot_ok(n_xm) ? (int32_t)m.esdf[(size_t)n_xm * 512 + src * 64 + 8 * vy + 7].x : PROP_NONE;
    if (vx == 7) v_xp = slot_ok(n_xp) ? (int32_t)m.esdf[(size_t)n_xp * 512 + src * 64 + 8 * vy + 0].x : PROP_NONE;
    if (vy == 0) v_ym = slot_ok(n_ym) ? (int32_t)m.esdf[(size_t)n_ym * 512 + src * 64 + 8 * 7 + vx].x : PROP_NONE;
    if (vy == 7) v_yp = slot_ok(n_yp) ? (int32_t)m.esdf[(size_t)n_yp * 512 + src * 64 + 8 * 0 + vx].x : PROP_NONE;
    int32_t best = own;
    best = min(best, prop_candidate(v_xm, -1, 0, a.max_sq)); best = min(best, prop_candidate(v_xp, 1, 0, a.max_sq));
    best = min(best, prop_candidate(v_ym, 0, -1, a.max_sq)); best = min(best, prop_candidate(v_yp, 0, 1, a.max_sq));
    m.esdf[(size_t)s * 512 + dst * 64 + lane] = make_uint2((uint32_t)best, 0u);
    if (__ballot(best != own) != 0ull && lane == 0) atomicOr(changed, 1);
  }
}
__global__ __launch_bounds__(64) void k_esdf_prop_final(DMap m, EsdfArgs a, int32_t src, int32_t other) {
  const int lane = threadIdx.x;
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x; s < hw; s += gridDim.x) {
    uint32_t flags;
    if (!prop_block(m, a, s, &flags)) continue;
    const int32_t p = (int32_t)m.esdf[(size_t)s * 512 + src * 64 + lane].x;
    const uint32_t vflags = (((m.obs_bits[s] >> lane) & 1ull) ? ESDF_OBSERVED : 0u) | (((m.inside_bits[s] >> lane) & 1ull) ? ESDF_INSIDE : 0u) |
                            (((m.site_bits[s] >> lane) & 1ull) ? ESDF_SITE : 0u);
    m.esdf[(size_t)s * 512 + src * 64 + lane] = make_uint2(0u, 0u);          // the scratch planes leave as they were: zero
    m.esdf[(size_t)s * 512 + other * 64 + lane] = make_uint2(0u, 0u);
    uint2 out = make_uint2(__float_as_uint(a.max_sq), vflags);
    if (p != PROP_NONE) {
      const int32_t fdx = (p & 127) - 64, fdy = ((p >> 7) & 127) - 64;
      out = make_uint2(__float_as_uint((float)(p >> 14)), vflags | ((uint32_t)(uint8_t)(int8_t)fdx) | (((uint32_t)(uint8_t)(int8_t)fdy) << 8));
    }
    m.esdf[(size_t)s * 512 + a.vz_out * 64 + lane] = out;
  }
}
static int run_esdf_propagation(nvbx_mapper* m, const EsdfArgs& a) {
  const int32_t pa = (a.vz_out + 1) & 7, pb = (a.vz_out + 2) & 7;
  const unsigned grid = (unsigned)std::min<int64_t>(m->capacity, 2048);
  int32_t* flag = m->export_count;                 // 64-byte device scratch
  NVBX_LAUNCH(m, k_esdf_prop_init, dim3(grid), dim3(64), m->d, a, pa);
  int32_t src = pa, dst = pb;
  for (int rounds = 0; rounds < (1 << 14); rounds += PROP_BATCH) {
    NVBX_HIP(hipMemsetAsync(flag, 0, 4, m->stream));
    for (int k = 0; k < PROP_BATCH; k++) { NVBX_LAUNCH(m, k_esdf_prop_iter, dim3(grid), dim3(64), m->d, a, src, dst, flag); std::swap(src, dst); }
    int32_t h = 0;
    NVBX_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, m->stream));
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (!h) break;                                   // a whole batch without a change: fixed point (PROP_BATCH is even: src == pa again)
  }
  NVBX_LAUNCH(m, k_esdf_prop_final, dim3(grid), dim3(64), m->d, a, src, dst);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

extern "C" int nvbx_update_esdf(nvbx_mapper* m) {
  if (!m) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  if (m->p.esdf_mode == 1) {                       // EsdfMode::k3D: esdf3d.hip
    if (m->p.esdf_max_distance_m / m->p.voxel_size >= 64.0f) { set_error("esdf_max_distance_m / voxel_size must be < 64 voxels"); return NVBX_E_INVALID; }
    if (m->join_side()) return NVBX_E_DEVICE;
    return m->update_esdf_3d();
  }
  // (parameters are validated before any state of this update changes)
  if (m->p.esdf_max_distance_m / m->p.voxel_size >= 64.0f) { set_error("esdf_max_distance_m / voxel_size must be < 64 voxels"); return NVBX_E_INVALID; }
  { const EsdfArgs chk = m->make_esdf_args();
    if (chk.bz_hi - chk.bz_lo + 1 > 63 || chk.bz_hi < chk.bz_lo) { set_error("esdf slice z band must span 1..63 blocks"); return NVBX_E_INVALID; } }
  // colour deferral (nvbx_mapper_set_color_deferral): while a colour frame is held back this update is held back behind it -- its
  // marking pass rides in that colour launch; both are carried out by the next integrateDepth (pipelined) or by whatever entry point
  // comes first (join_side -> replay_deferred, in call order)
  // (... or, with the switch on and no colour frame around, behind nothing: the next camera depth frame carries it alone -- depth-only hosts,
  //  occupancy mappers; nvbx_mapper::esdf_only_carry)
  static const int esdf_only = getenv("NVBX_ESDF_ONLY_CARRY") ? atoi(getenv("NVBX_ESDF_ONLY_CARRY")) : 1;      // (A/B: 0 = an updateEsdf is held back behind a colour frame only)
  if ((m->color_pending.on || (esdf_only && m->color_deferral && !m->pipelined_order && !m->import_pending)) && !m->replaying && m->p.esdf_propagation == 0 && !m->use_side && m->defer_edt) {
    // (an update already held back is carried out first: two updates in a row stay two updates -- epochs and the "last update" counters as in
    //  the undeferred sequence)
    if (m->esdf_update_pending && m->replay_deferred()) return NVBX_E_DEVICE;
    if (m->color_pending.on || (esdf_only && m->color_deferral && !m->import_pending)) { m->esdf_update_pending = true; return NVBX_OK; }
  }
  if (!m->replaying && !m->pipelined_order && m->replay_deferred()) return NVBX_E_DEVICE;
  // a distance transform still held back by the PREVIOUS update goes first: this update's marking pass overwrites the masks
  // and the parity-indexed window record it reads, and edt_args holds one update only (two updates back to back)
  if (m->flush_edt()) return NVBX_E_DEVICE;
  // a held-back union step belongs to this update -- except in pipelined order with the fused launch, where it rides in the TSDF-update launch
  // that follows and dirties the peers' blocks for the NEXT marking pass (tsdf.hip)
  if (!m->pipelined_order && m->flush_import()) return NVBX_E_DEVICE;
  if (m->dirty_since_mark) m->mark_pass++;
  const EsdfArgs a = m->make_esdf_args();
  hipStream_t s = m->stream;
  if (m->p.esdf_propagation == 1) {                // [U] open choice: iterative propagation instead of the exact transform (above)
    if (m->join_side()) return NVBX_E_DEVICE;
    if (m->dirty_since_mark) NVBX_LAUNCH(m, k_esdf_mark, dim3((unsigned)std::min<int64_t>(m->capacity, 1024)), dim3(64), m->d, a);
    m->dirty_since_mark = false; m->premark_consumed = false; m->unresolved_marks = false; m->pass_at_last_edt = m->mark_pass;
    const int rc = run_esdf_propagation(m, a);
    m->esdf_epoch++;
    return rc;
  }
  if (m->use_side) {
    // run behind the last non-colour operation, beside any colour integration enqueued after it
    if (m->main_dirty) { if (m->mark_main()) return NVBX_E_DEVICE; }
    NVBX_HIP(hipStreamWaitEvent(m->side, m->ev_main, 0));
    s = m->side;
  }
  // marking pass only if something was dirtied since the last pass (integrateColor runs one fused into its own launch)
  if (m->dirty_since_mark) NVBX_LAUNCH_ON(m, s, k_esdf_mark, dim3((unsigned)std::min<int64_t>(m->capacity, 1024)), dim3(64), m->d, a);
  m->dirty_since_mark = false; m->premark_consumed = false;          // k_esdf_edt resets the dirty list
  m->unresolved_marks = false; m->pass_at_last_edt = m->mark_pass;    // the EDT enqueued below resolves every marking pass so far
  if (m->defer_edt && !m->use_side) {
    // The EDT is held back until the next entry point: the next camera depth frame runs it inside its first launch (beside
    // the view marking, which it does not interact with); every other entry point launches it first (flush_edt).
    m->edt_pending = true; m->edt_args = a;
  } else {
    NVBX_LAUNCH_ON(m, s, k_esdf_edt, dim3(1024), dim3(256), m->d, a);
  }
  NVBX_HIP(hipGetLastError());
  if (m->use_side) { NVBX_HIP(hipEventRecord(m->ev_side, m->side)); m->side_pending = true; }
  m->esdf_epoch++;
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ slicer
__global__ __launch_bounds__(64) void k_esdf_slice(DMap m, int32_t bz_out, int32_t vz_out, float voxel_size, float unknown, float* img,
                                                   int32_t bx0, int32_t by0, int32_t nbx, int32_t nby) {
  const int lane = threadIdx.x;
  const int vx = lane & 7, vy = lane >> 3;
  const int32_t cols = nbx * 8;
  for (int32_t c = blockIdx.x; c < nbx * nby; c += gridDim.x) {
    const int32_t cy = c / nbx, cx = c - cy * nbx;
    const uint32_t es = find_slot(m, bx0 + cx, by0 + cy, bz_out, F_ESDF);
    float v = unknown;
    if (slot_ok(es)) {
      const uint2 e = m.esdf[(size_t)es * 512 + vz_out * 64 + lane];
      if (e.y & ESDF_OBSERVED) { v = sqrtf(__uint_as_float(e.x)) * voxel_size; if (e.y & ESDF_INSIDE) v = -v; }
    }
    img[(int64_t)(cy * 8 + vy) * cols + cx * 8 + vx] = v;
  }
}

extern "C" int nvbx_esdf_slice_size(nvbx_mapper* m, int32_t* rows, int32_t* cols, float aabb[6]) {
  if (!m || !rows || !cols) return NVBX_E_INVALID;
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  m->slice_size_seq = m->enqueue_seq;          // (nvbx_esdf_slice_to_image right behind this call needs no second fetch)
  const int32_t* c = m->h_counters + C_ESDF_AABB;
  if (c[0] > c[2]) { *rows = 0; *cols = 0; return NVBX_OK; }
  const EsdfArgs a = m->make_esdf_args();
  const float bs = m->p.voxel_size * 8.0f;
  *cols = (c[2] - c[0] + 1) * 8; *rows = (c[3] - c[1] + 1) * 8;
  if (aabb) {
    aabb[0] = (float)c[0] * bs; aabb[1] = (float)c[1] * bs; aabb[2] = (float)a.bz_out * bs;
    aabb[3] = (float)(c[2] + 1) * bs; aabb[4] = (float)(c[3] + 1) * bs; aabb[5] = (float)(a.bz_out + 1) * bs;
  }
  return NVBX_OK;
}

extern "C" int nvbx_esdf_slice_to_image(nvbx_mapper* m, float unknown_value, float* image_dev, int64_t capacity_elems, int32_t* rows,
                                        int32_t* cols, float aabb[6]) {
  if (!m || !rows || !cols) return NVBX_E_INVALID;
  // (called right behind nvbx_esdf_slice_size -- the facade's EsdfSlicer sizes its image first -- nothing has been enqueued since: the counters
  //  on the host are still the device's, no second round trip)
  int rc = NVBX_OK;
  if (m->slice_size_seq != 0 && m->slice_size_seq == m->enqueue_seq && !m->color_pending.on && !m->esdf_update_pending && !m->edt_pending && !m->import_pending && !m->side_pending) {
    // (the parts of join_side() that still apply: this mapper's device is the current one -- the call may come from another thread, or behind a call
    //  into a mapper on another GPU -- and the launch below counts as work on the main stream; nothing is held back and the side stream is joined)
    { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != m->device) NVBX_HIP(hipSetDevice(m->device)); }
    m->main_dirty = true;
    const int32_t* c0 = m->h_counters + C_ESDF_AABB;
    if (c0[0] > c0[2]) { *rows = 0; *cols = 0; } else { *cols = (c0[2] - c0[0] + 1) * 8; *rows = (c0[3] - c0[1] + 1) * 8; }
    if (aabb && *rows) {
      const EsdfArgs a0 = m->make_esdf_args(); const float bs = m->p.voxel_size * 8.0f;
      aabb[0] = (float)c0[0] * bs; aabb[1] = (float)c0[1] * bs; aabb[2] = (float)a0.bz_out * bs;
      aabb[3] = (float)(c0[2] + 1) * bs; aabb[4] = (float)(c0[3] + 1) * bs; aabb[5] = (float)(a0.bz_out + 1) * bs;
    }
  } else { rc = nvbx_esdf_slice_size(m, rows, cols, aabb); if (rc) return rc; }
  if (*rows == 0) return NVBX_OK;
  if (!image_dev || (int64_t)*rows * *cols > capacity_elems) { set_error("slice image capacity too small"); return NVBX_E_CAPACITY; }
  const EsdfArgs a = m->make_esdf_args();
  const int32_t* c = m->h_counters + C_ESDF_AABB;
  const int32_t nbx = c[2] - c[0] + 1, nby = c[3] - c[1] + 1;
  NVBX_LAUNCH(m, k_esdf_slice, dim3(std::min(nbx * nby, 4096)), dim3(64), m->d, a.bz_out, a.vz_out, m->p.voxel_size,
                     unknown_value, image_dev, c[0], c[1], nbx, nby);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

static int ensure_staging(nvbx_mapper* m, int64_t bytes) {
  if (bytes <= m->staging_bytes) return NVBX_OK;
  NVBX_HIP(hipStreamSynchronize(m->stream));
  if (m->staging) NVBX_HIP(hipFree(m->staging));
  m->staging = nullptr; m->staging_bytes = 0;
  NVBX_HIP(hipMalloc(&m->staging, bytes));
  m->staging_bytes = bytes;
  return NVBX_OK;
}

// The slice for a HOST caller in ONE wait (processEsdf slices right after updateEsdf and publishes from the host, nvblox_node.cpp:774-889: ten
// times a second the node drains its pipeline here).  Size, then image, then download used to be three round trips to the device -- the layer's AABB
// had to reach the host before the slicing launch could be sized.  k_esdf_slice_rows reads the AABB itself (device counters), writes
// {min_x, min_y, max_x, max_y, status} + the image straight into pinned host memory, in whole 256-B row segments (one wavefront per strip of 8
// blocks: lane = pixel of a 64-pixel row segment), and the host waits once.  status: 1 = image written, 0 = no ESDF block, -1 = `cap` too small.
__global__ __launch_bounds__(64) void k_esdf_slice_rows(DMap m, int32_t bz_out, int32_t vz_out, float voxel_size, float unknown, float* img, int64_t cap, int32_t* header) {
  const int32_t bx0 = m.counters[C_ESDF_AABB], by0 = m.counters[C_ESDF_AABB + 1], bx1 = m.counters[C_ESDF_AABB + 2], by1 = m.counters[C_ESDF_AABB + 3];
  const int lane = (int)threadIdx.x;
  const bool empty = bx0 > bx1;
  const int32_t nbx = empty ? 0 : bx1 - bx0 + 1, nby = empty ? 0 : by1 - by0 + 1;
  const bool fits = (int64_t)nbx * nby * 64 <= cap;
  if (blockIdx.x == 0 && lane < 5) header[lane] = lane == 0 ? bx0 : lane == 1 ? by0 : lane == 2 ? bx1 : lane == 3 ? by1 : (empty ? 0 : (fits ? 1 : -1));
  if (empty || !fits) return;
  const int32_t strips_x = (nbx + 7) / 8, cols = nbx * 8;
  for (int32_t s = (int32_t)blockIdx.x; s < strips_x * nby; s += (int32_t)gridDim.x) {
    const int32_t cy = s / strips_x, cx0 = (s - cy * strips_x) * 8;
    // lanes 0..7 look the strip's blocks up; lane l then reads pixel (l & 7) of block (l >> 3), row r
    uint32_t es = SLOT_NONE;
    if (lane < 8 && cx0 + lane < nbx) es = find_slot(m, bx0 + cx0 + lane, by0 + cy, bz_out, F_ESDF);
    const uint32_t mine = __shfl(es, lane >> 3);
    const bool in_img = cx0 + (lane >> 3) < nbx;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      float v = unknown;
      if (slot_ok(mine)) {
        const uint2 e = m.esdf[(size_t)mine * 512 + vz_out * 64 + r * 8 + (lane & 7)];
        if (e.y & ESDF_OBSERVED) { v = sqrtf(__uint_as_float(e.x)) * voxel_size; if (e.y & ESDF_INSIDE) v = -v; }
      }
      if (in_img) img[(int64_t)(cy * 8 + r) * cols + cx0 * 8 + lane] = v;
    }
  }
}
static int ensure_slice_pinned(nvbx_mapper* m, int64_t elems) {
  if (elems <= m->slice_pinned_elems) return NVBX_OK;
  NVBX_HIP(hipStreamSynchronize(m->stream));
  if (m->slice_pinned) NVBX_HIP(hipHostFree(m->slice_pinned));
  m->slice_pinned = nullptr; m->slice_pinned_elems = 0; m->slice_pinned_dev = nullptr;
  NVBX_HIP(hipHostMalloc((void**)&m->slice_pinned, (size_t)(elems + 16) * 4, hipHostMallocMapped));
  { void* dp = nullptr; NVBX_HIP(hipHostGetDevicePointer(&dp, m->slice_pinned, 0)); m->slice_pinned_dev = (float*)dp; }
  m->slice_pinned_elems = elems;
  return NVBX_OK;
}

extern "C" int nvbx_esdf_slice_to_host(nvbx_mapper* m, float unknown_value, float* image_host, int64_t capacity_elems, int32_t* rows,
                                       int32_t* cols, float aabb[6]) {
  if (!m || !rows || !cols) return NVBX_E_INVALID;
  *rows = 0; *cols = 0;
  if (m->join_side()) return NVBX_E_DEVICE;
  // (the pinned image is at least as large as the caller's, and never smaller than 256 x 256: a caller that passes no buffer learns the size)
  int rc = ensure_slice_pinned(m, std::max<int64_t>(image_host ? capacity_elems : 0, 65536)); if (rc) return rc;
  const EsdfArgs a = m->make_esdf_args();
  int32_t* header = reinterpret_cast<int32_t*>(m->slice_pinned_dev);          // [0..15] header, image behind it
  volatile int32_t* h = reinterpret_cast<volatile int32_t*>(m->slice_pinned);
  NVBX_LAUNCH(m, k_esdf_slice_rows, dim3(1024), dim3(64), m->d, a.bz_out, a.vz_out, m->p.voxel_size, unknown_value, m->slice_pinned_dev + 16, m->slice_pinned_elems, header);
  NVBX_HIP(hipGetLastError());
  if (m->wait_stream()) return NVBX_E_DEVICE;
  if (h[4] == 0) return NVBX_OK;
  const float bs = m->p.voxel_size * 8.0f;
  *cols = (h[2] - h[0] + 1) * 8; *rows = (h[3] - h[1] + 1) * 8;
  if (aabb) {
    aabb[0] = (float)h[0] * bs; aabb[1] = (float)h[1] * bs; aabb[2] = (float)a.bz_out * bs;
    aabb[3] = (float)(h[2] + 1) * bs; aabb[4] = (float)(h[3] + 1) * bs; aabb[5] = (float)(a.bz_out + 1) * bs;
  }
  const int64_t n = (int64_t)*rows * *cols;
  if (!image_host || n > capacity_elems) { set_error("slice image capacity too small"); return NVBX_E_CAPACITY; }
  if (h[4] < 0) {          // larger than the pinned image (the layer grew past it between two calls): once more with room
    rc = ensure_slice_pinned(m, n + n / 2); if (rc) return rc;
    return nvbx_esdf_slice_to_host(m, unknown_value, image_host, capacity_elems, rows, cols, aabb);
  }
  memcpy(image_host, m->slice_pinned + 16, (size_t)n * 4);
  return NVBX_OK;
}

__global__ void k_occupancy(const float* img, int64_t n, float unknown, int8_t* grid) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = img[i];
    int8_t o = 0;
    if (fabsf(v - unknown) < 1e-2f) o = -1; else if (v <= 0.0f) o = 100;
    grid[i] = o;
  }
}
extern "C" int nvbx_occupancy_grid_from_slice(nvbx_mapper* m, const float* image_dev, int32_t rows, int32_t cols, float unknown_value,
                                              int8_t* grid_dev) {
  if (!m || !image_dev || !grid_dev || rows <= 0 || cols <= 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  const int64_t n = (int64_t)rows * cols;
  NVBX_LAUNCH(m, k_occupancy, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), image_dev, n, unknown_value, grid_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// conversions/esdf_slice_conversions.cu:33-73 restated for wave64: one wave per 8x8 pixel tile, one atomicAdd per
// wave (ballot + popcount prefix) instead of one per pixel.
__global__ __launch_bounds__(64) void k_slice_pointcloud(DMap m, const float* img, int32_t rows, int32_t cols, float ax, float ay,
                                                         float slice_height, float voxel_size, float unknown, float4* out) {
  const int lane = threadIdx.x;
  const int tiles_x = (cols + 7) >> 3, tiles_y = (rows + 7) >> 3;
  for (int32_t t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int r = ty * 8 + (lane >> 3), c = tx * 8 + (lane & 7);
    bool keep = false; float v = 0.0f;
    if (r < rows && c < cols) { v = img[(int64_t)r * cols + c]; keep = !(fabsf(v - unknown) < 1e-2f); }
    const u64 mask = __ballot(keep);
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(&m.counters[C_TMP], (int)__popcll(mask));
    base = __shfl(base, 0);
    if (keep) {
      const int off = (int)__popcll(mask & ((1ull << lane) - 1ull));
      out[base + off] = make_float4(ax + voxel_size * (float)c, ay + voxel_size * (float)r, slice_height, v);
    }
  }
}
__global__ void k_zero_tmp2(DMap m) { m.counters[C_TMP] = 0; }

extern "C" int nvbx_pointcloud_from_slice(nvbx_mapper* m, const float* image_dev, int32_t rows, int32_t cols, const float aabb[6],
                                          float slice_height, float unknown_value, float* points_xyzi_dev, int32_t* n_points) {
  if (!m || !image_dev || !aabb || !points_xyzi_dev || !n_points || rows <= 0 || cols <= 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  NVBX_LAUNCH(m, k_zero_tmp2, dim3(1), dim3(1), m->d);
  const int tiles = ((rows + 7) / 8) * ((cols + 7) / 8);
  NVBX_LAUNCH(m, k_slice_pointcloud, dim3(std::min(tiles, 4096)), dim3(64), m->d, image_dev, rows, cols, aabb[0], aabb[1],
                     slice_height, m->p.voxel_size, unknown_value, (float4*)points_xyzi_dev);
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  *n_points = m->h_counters[C_TMP];
  return NVBX_OK;
}

// voxelLayerToDenseVoxelGridInAABBAsync<SignedDistanceFunctor> (esdf_and_gradients_conversions.cu:28-48,88-125)
__global__ void k_esdf_dense(DMap m, int32_t mx, int32_t my, int32_t mz, int32_t sx, int32_t sy, int32_t sz, float voxel_size, float def, float* out) {
  const int64_t n = (int64_t)sx * sy * sz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t z = (int32_t)(i % sz), y = (int32_t)((i / sz) % sy), x = (int32_t)(i / ((int64_t)sz * sy));
    const int32_t gx = mx + x, gy = my + y, gz = mz + z;
    const uint32_t es = find_slot(m, gx >> 3, gy >> 3, gz >> 3, F_ESDF);
    float v = def;
    if (slot_ok(es)) {
      const uint2 e = m.esdf[(size_t)es * 512 + (gx & 7) + 8 * (gy & 7) + 64 * (gz & 7)];
      if (e.y & ESDF_OBSERVED) { v = sqrtf(__uint_as_float(e.x)) * voxel_size; if (e.y & ESDF_INSIDE) v = v * -1.0f; }
    }
    out[i] = v;
  }
}
extern "C" int nvbx_esdf_dense_grid(nvbx_mapper* m, const int32_t min_vox[3], const int32_t size_vox[3], float default_value, float* grid_dev) {
  if (!m || !min_vox || !size_vox || !grid_dev || size_vox[0] <= 0 || size_vox[1] <= 0 || size_vox[2] <= 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  const int64_t n = (int64_t)size_vox[0] * size_vox[1] * size_vox[2];
  NVBX_LAUNCH(m, k_esdf_dense, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), m->d, min_vox[0], min_vox[1],
                     min_vox[2], size_vox[0], size_vox[1], size_vox[2], m->p.voxel_size, default_value, grid_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU hooks
__global__ void k_export_dirty(DMap m, int32_t* out_idx, int32_t* out_count, int32_t cap) {
  ListView lv;
  int32_t n = list_open(m, S_LIST_ESDF_DIRTY, &lv); if (n > cap) n = cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = n;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t s = list_at(m, S_LIST_ESDF_DIRTY, lv, i);
    out_idx[3 * i] = m.slot_index[3 * s]; out_idx[3 * i + 1] = m.slot_index[3 * s + 1]; out_idx[3 * i + 2] = m.slot_index[3 * s + 2];
  }
}
extern "C" int nvbx_esdf_dirty_list(nvbx_mapper* m, int32_t* indices_dev_out, int32_t* count_dev_out, int64_t capacity) {
  if (!m || !indices_dev_out || !count_dev_out || capacity <= 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  // (a list already consumed by a marking pass still names the blocks dirtied since the last updateEsdf: export it as is)
  NVBX_LAUNCH(m, k_export_dirty, dim3(64), dim3(256), m->d, indices_dev_out, count_dev_out, (int32_t)std::min<int64_t>(capacity, m->capacity));
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
// Registration only (no launch): every following integrateDepth writes the Index3D of the blocks it updates, packed as
// int32 [1 + capacity][3] with row 0 = {count, 0, 0}, into this caller-owned device buffer (NULL: off).
extern "C" int nvbx_set_view_export(nvbx_mapper* m, int32_t* packed_dev, int64_t capacity) {
  if (!m || capacity < 0) return NVBX_E_INVALID;
  m->view_export = packed_dev; m->view_export_cap = packed_dev ? std::min<int64_t>(capacity, m->capacity) : 0;
  return NVBX_OK;
}
// Union step after the all-gather: blocks another GPU updated that exist locally as TSDF blocks become ESDF-dirty here.
__global__ void k_import_dirty(DMap m, const int32_t* idx, const int32_t* count, int64_t max_count) {
  int64_t n = *count; if (n > max_count) n = max_count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], F_TSDF);
    if (!slot_ok(s)) continue;
    const uint32_t old = atomicOr(&m.slot_flags[s], F_DIRTY_ESDF);
    if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)s);
  }
}
extern "C" int nvbx_mark_esdf_dirty(nvbx_mapper* m, const int32_t* indices_dev, const int32_t* count_dev, int64_t max_count) {
  if (!m || !indices_dev || !count_dev || max_count < 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  NVBX_LAUNCH(m, k_import_dirty, dim3(64), dim3(256), m->d, indices_dev, count_dev, max_count);
  NVBX_HIP(hipGetLastError());
  return m->mark_main();
}

// all peers' lists in one launch (blockIdx.y = rank)
__global__ void k_import_dirty_gathered(DMap m, const int32_t* g, int32_t self_rank, int64_t max_count) {
  const int32_t r = (int32_t)blockIdx.y;
  if (r == self_rank) return;
  const int32_t* base = g + (size_t)r * (size_t)(max_count + 1) * 3;
  int64_t n = base[0]; if (n > max_count) n = max_count;
  const int32_t* idx = base + 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], F_TSDF);
    if (!slot_ok(s)) continue;
    const uint32_t old = atomicOr(&m.slot_flags[s], F_DIRTY_ESDF);
    if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)s);
  }
}
// 2-D ESDF: lookup AND site marking of every peer's blocks in one launch (one wavefront per list entry, blockIdx.y = rank): the
// block is re-marked on the spot, exactly as a marking pass would after finding it on the dirty list, so the ESDF update that
// follows needs no marking launch of its own for the peers' lists (and no list reset before them).
__global__ __launch_bounds__(64) void k_import_mark_gathered(DMap m, EsdfArgs a, const int32_t* g, int32_t self_rank, int64_t max_count) {
  const int32_t r = (int32_t)blockIdx.y;
  if (r == self_rank) return;
  const int32_t* base = g + (size_t)r * (size_t)(max_count + 1) * 3;
  int64_t n = base[0]; if (n > max_count) n = max_count;
  const int32_t* idx = base + 3;
  const int srec = S_ESDF_REC + (int)(a.epoch & 1), sh = my_shard();
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], F_TSDF);        // (uniform: every lane asks the same)
    if (!slot_ok(s)) continue;
    esdf_mark_entry(m, a, s, srec, sh);
  }
}
extern "C" int nvbx_mark_esdf_dirty_gathered(nvbx_mapper* m, const int32_t* gathered_dev, int32_t world, int32_t self_rank, int64_t max_count) {
  if (!m || !gathered_dev || world < 1 || max_count < 0) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;
  if (!(world > 1 || self_rank < 0)) return NVBX_OK;
  const EsdfArgs probe = m->make_esdf_args();
  if (m->p.esdf_mode == 0 && probe.bz_hi >= probe.bz_lo && probe.bz_hi - probe.bz_lo + 1 <= 63) {
    m->mark_pass++;
    const EsdfArgs a = m->make_esdf_args();
    m->unresolved_marks = true;                    // (a marking pass that no distance transform has followed yet)
    NVBX_LAUNCH(m, k_import_mark_gathered, dim3(256, (unsigned)world), dim3(64), m->d, a, gathered_dev, self_rank, max_count);
  } else {
    if (m->begin_dirtying()) return NVBX_E_DEVICE;
    NVBX_LAUNCH(m, k_import_dirty_gathered, dim3(16, (unsigned)world), dim3(256), m->d, gathered_dev, self_rank, max_count);
  }
  NVBX_HIP(hipGetLastError());
  return m->mark_main();
}

// Take back the marking passes that no distance transform has followed yet (nvbx_mapper.h: unresolved_marks).  A marking pass
// consumes ESDF-dirty flags and creates PENDING columns from the TSDF as it is at that moment; if blocks are deallocated
// before the update runs, the update must start from the dirty set and the TSDF of THAT moment instead.  One wavefront per
// slot: (a) a TSDF slot whose dirty flag an unresolved pass consumed is dirty again (and back on the list); (b) an ESDF
// column an unresolved pass re-marked gets the masks of its voxels back (what the last distance transform wrote), a column
// that was only PENDING disappears.  Slots freed here are unlinked by the hash rebuild of the calling operation.
__global__ __launch_bounds__(64) void k_undo_marks(DMap m, uint32_t pass_floor, int32_t srec, int32_t vz_out) {
  const int lane = threadIdx.x;
  // (the update's window record keeps the undone passes' extent: the distance transform is exact on any window that
  // contains the changes, and the record may also hold the extent of ESDF blocks dropped by clearOutsideRadius)
  if (blockIdx.x == 0 && lane < NSH) *shc_at(m, srec, lane, 4) = 0;     // columns re-marked: counted again by the pass that follows
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (int32_t s = blockIdx.x; s < hw; s += gridDim.x) {
    const uint32_t flags = m.slot_flags[s], cons = m.slot_consumed[s], stamp = m.slot_stamp[s];
    const bool cons_unres = cons != STAMP_NEVER && (int32_t)(cons - pass_floor) > 0;
    const bool stamp_unres = stamp != STAMP_NEVER && (int32_t)(stamp - pass_floor) > 0;
    if (cons_unres && lane == 0) {
      m.slot_consumed[s] = STAMP_NEVER;
      if (flags & (F_TSDF | F_COLOR | F_ESDF | F_MESH | F_ESDF_PENDING)) {
        const uint32_t old = atomicOr(&m.slot_flags[s], F_DIRTY_ESDF);
        if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, s);
      }
    }
    if (stamp_unres) {
      if (flags & F_ESDF) {
        const uint32_t meta = m.esdf[(size_t)s * 512 + vz_out * 64 + lane].y;
        const u64 sb = __ballot((meta & ESDF_SITE) != 0), ob = __ballot((meta & ESDF_OBSERVED) != 0), ib = __ballot((meta & ESDF_INSIDE) != 0);
        if (lane == 0) { m.site_bits[s] = sb; m.obs_bits[s] = ob; m.inside_bits[s] = ib; m.slot_stamp[s] = STAMP_NEVER; }
      } else if ((flags & F_ESDF_PENDING) && lane == 0) {
        atomicAnd(&m.slot_flags[s], ~F_ESDF_PENDING);
        m.site_bits[s] = 0ull; m.obs_bits[s] = 0ull; m.inside_bits[s] = 0ull; m.slot_stamp[s] = STAMP_NEVER;
        if (!(flags & (F_TSDF | F_COLOR | F_ESDF | F_MESH))) {
          atomicAnd(&m.slot_flags[s], ~(F_DIRTY_ESDF | F_DIRTY_MESH));
          const int32_t pos = atomicAdd(&m.counters[C_FREE_TOP], 1);
          m.free_stack[pos] = (uint32_t)s;

        }
      }
    }
  }
}
int nvbx_mapper::undo_marks() {
  if (!unresolved_marks) return NVBX_OK;
  if (reset_consumed_list()) return NVBX_E_DEVICE;           // a consumed list is emptied; the kernel re-appends what is dirty again
  const EsdfArgs a = make_esdf_args();
  NVBX_LAUNCH(this, k_undo_marks, dim3((unsigned)std::min<int64_t>(capacity, 2048)), dim3(64), d, pass_at_last_edt,
              (int32_t)(S_ESDF_REC + (int)(a.epoch & 1)), a.vz_out);
  NVBX_HIP(hipGetLastError());
  unresolved_marks = false; dirty_since_mark = true;
  return NVBX_OK;
}

// Deferred form of the union step: remembered, and performed by extra workgroups of the next integrateColor launch (beside the
// marking of the mapper's own dirty blocks, same marking pass) -- or by flush_import() if any other entry point comes first.
// The gathered buffer must stay valid and unchanged until then.
extern "C" int nvbx_mark_esdf_dirty_gathered_deferred(nvbx_mapper* m, const int32_t* gathered_dev, int32_t world, int32_t self_rank, int64_t max_count) {
  if (!m || !gathered_dev || world < 1 || max_count < 0) return NVBX_E_INVALID;
  if (m->flush_import()) return NVBX_E_DEVICE;                   // at most one held-back union step
  if (!(world > 1 || self_rank < 0)) return NVBX_OK;
  const EsdfArgs probe = m->make_esdf_args();
  if (!(m->p.esdf_mode == 0 && probe.bz_hi >= probe.bz_lo && probe.bz_hi - probe.bz_lo + 1 <= 63))
    return nvbx_mark_esdf_dirty_gathered(m, gathered_dev, world, self_rank, max_count);
  m->import_pending = true; m->import_ptr = gathered_dev; m->import_world = world; m->import_rank = self_rank; m->import_max = max_count;
  return NVBX_OK;
}
int nvbx_mapper::flush_import() {
  if (!import_pending) return NVBX_OK;
  import_pending = false;
  if (flush_edt()) return NVBX_E_DEVICE;                         // the EDT of the previous update reads the masks this marking overwrites
  mark_pass++;
  const EsdfArgs a = make_esdf_args();
  unresolved_marks = true;
  NVBX_LAUNCH(this, k_import_mark_gathered, dim3(256, (unsigned)import_world), dim3(64), d, a, import_ptr, import_rank, import_max);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// launch a held-back EDT now (every entry point except the camera integrateDepth, which lets it ride in k_mark_view)
int nvbx_mapper::flush_edt() {
  if (edt_pending) {
    edt_pending = false;
    NVBX_LAUNCH(this, k_esdf_edt, dim3(1024), dim3(256), d, edt_args);
    NVBX_HIP(hipGetLastError());
  }
  return NVBX_OK;
}
