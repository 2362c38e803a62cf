// nvbx_esdf_mark.h -- the ESDF site-marking worker (first half of MultiMapper::updateEsdf), shared by k_esdf_mark (esdf.hip)
// and by k_integrate_color (color.hip), which runs it in extra workgroups of its own launch: marking depends only on the
// TSDF, exactly like colour integration, so the two overlap inside one launch and a following updateEsdf needs the EDT only.
#pragma once
#include "nvbx_mapper.h"

namespace nvbx {

// Dependent-access chain: {shard counts of the dirty list} -> {dirty slot} -> {flags, Index3D} -> {hash entries of the ESDF block and of the
// TSDF z-band blocks, one per lane, in flight together} -> {column stamp exchange || TSDF column loads} -> store.
// One list entry (one wavefront): re-mark the ESDF column of TSDF slot `tslot` (or of the ESDF slot itself if it is flagged REMARK).
__device__ inline void esdf_mark_entry(const DMap& m, const EsdfArgs& a, uint32_t tslot, int srec, int sh, bool from_dirty_list = false) {
  const int lane = threadIdx.x & 63;
  const int vx = lane & 7, vy = lane >> 3;
  const uint32_t tflags = m.slot_flags[tslot];
  const int32_t bx = m.slot_index[3 * tslot], by = m.slot_index[3 * tslot + 1], bz = m.slot_index[3 * tslot + 2];
  // the z band: fixed heights, or -- [U] ground-plane mode -- relative to the plane: per COLUMN kz_lo .. kz_hi from the plane's height at the
  // column's centre; the blocks the column's wavefront may have to read span the heights at the block's four corners (a plane is extremal there)
  int32_t kz_lo = a.kz_min, kz_hi = a.kz_max, bz_lo = a.bz_lo, bz_hi = a.bz_hi;
  if (a.plane_on) {
    const float bs = a.voxel_size * 8.0f;
    const float hl = esdf_plane_height(a.pl, voxel_center(bx, vx, bs, a.voxel_size), voxel_center(by, vy, bs, a.voxel_size));
    kz_lo = (int32_t)floorf((hl + a.above) / a.voxel_size); kz_hi = (int32_t)floorf(((hl + a.above) + a.thick) / a.voxel_size);
    const float x0 = (float)bx * bs, x1 = (float)(bx + 1) * bs, y0 = (float)by * bs, y1 = (float)(by + 1) * bs;
    const float h00 = esdf_plane_height(a.pl, x0, y0), h10 = esdf_plane_height(a.pl, x1, y0), h01 = esdf_plane_height(a.pl, x0, y1), h11 = esdf_plane_height(a.pl, x1, y1);
    const float hmin = fminf(fminf(h00, h10), fminf(h01, h11)), hmax = fmaxf(fmaxf(h00, h10), fmaxf(h01, h11));
    bz_lo = floor_div8((int32_t)floorf((hmin + a.above) / a.voxel_size)); bz_hi = floor_div8((int32_t)floorf(((hmax + a.above) + a.thick) / a.voxel_size));
    if (bz_hi - bz_lo > 61) bz_hi = bz_lo + 61;            // (a wavefront probes the ESDF block + at most 62 band blocks)
  }
  const int nz = bz_hi - bz_lo + 1;                        // TSDF blocks spanned by the slice z band (<= 62)
  // An entry of the ESDF-dirty list names a slot that carried F_DIRTY_ESDF when it was appended (the flag is the list's de-duplication); a slot
  // WITHOUT it has been freed since (decay, clearing) -- and may be handed out again at this very moment: the marking pass of a held-back update
  // rides in the view-marking launch of the NEXT frame, whose tiles allocate.  The new block's dirtiness belongs to the next
  // update (its TSDF update sets the flag one launch later), so the stale entry is skipped; found as ESDF columns the sequential order never
  // creates when the new block was deallocated again before that update (tests/test_gpu_sequences.py, seed 3).
  if (from_dirty_list && !(tflags & F_DIRTY_ESDF)) return;
  NVBX_INV_TSDF_READER(m);
  NVBX_INV_COUNT(m, C_INV_I4, lane == 0 && from_dirty_list && !(tflags & (F_TSDF | F_ESDF)));
  NVBX_TV(0, 3, wall_clock64());
  if (lane == 0) { atomicAnd(&m.slot_flags[tslot], ~F_DIRTY_ESDF); m.slot_consumed[tslot] = a.mark_pass; }
  // a dirty TSDF block of the z band dirties its column; an ESDF slot flagged F_ESDF_REMARK (a TSDF block of its band was
  // deallocated by decay) re-marks its own column
  if (!(tflags & F_ESDF_REMARK) && (bz < bz_lo || bz > bz_hi)) return;
  // lane 0: the ESDF block (x, y, z_slice); lanes 1..nz: the TSDF blocks of the band -- one probe each, together
  const int32_t qz = lane == 0 ? a.bz_out : bz_lo + lane - 1;
  const bool probing = lane <= nz;
  const u64 qkey = pack_key(bx, by, qz);
  const uint32_t qh = probing ? table_pos(m, bx, by, qz) : 0u;
  const uint4 qe = ld_entry(m, qh);
  uint32_t qslot = probing ? resolve_any(m, qkey, qh, qe) : SLOT_NONE;
  uint32_t eslot = __shfl(qslot, 0);
  NVBX_TV(0, 4, wall_clock64());
  // Everything that depends on the probes alone is requested TOGETHER, one round trip instead of three in a row: the ESDF slot's flags, the
  // column stamp exchange (the common case: the entry is a live TSDF block -- then the early return below cannot be taken -- and the column
  // exists), and the first two TSDF blocks of the band (the shipped configurations span two).
  const bool early = (tflags & F_TSDF) && slot_ok(eslot);
  int first = 0;
  if (early && lane == 0) first = atomicExch(&m.slot_stamp[eslot], a.mark_pass) != a.mark_pass;
  const uint32_t ts0 = __shfl(qslot, 1), ts1 = __shfl(qslot, nz > 1 ? 2 : 1);
  const float4* c0 = reinterpret_cast<const float4*>(&m.tsdf[(size_t)(slot_ok(ts0) ? ts0 : 0) * 512 + 64 * vx + 8 * vy]);
  const float4* c1 = reinterpret_cast<const float4*>(&m.tsdf[(size_t)(slot_ok(ts1) ? ts1 : 0) * 512 + 64 * vx + 8 * vy]);
  const float4 p00 = c0[0], p01 = c0[1], p02 = c0[2], p03 = c0[3], p10 = c1[0], p11 = c1[1], p12 = c1[2], p13 = c1[3];      // (named registers: an indexed array of them went to scratch memory)
  const bool e_exists = slot_ok(eslot) && (m.slot_flags[slot_ok(eslot) ? eslot : 0] & (F_ESDF | F_ESDF_PENDING));
  if (!(tflags & F_TSDF) && !(e_exists && (tflags & F_ESDF_REMARK))) return;          // uniform
  if (lane == 0) {
    if (!slot_ok(eslot)) {                                // new column: insert (device-side allocation)
      bool is_new;
      const int32_t h = hash_insert(m, bx, by, a.bz_out, F_ESDF_PENDING, &is_new);
      if (h >= 0) { do { eslot = ld_slot_acquire(&m.table[h]); } while (eslot == SLOT_INVALID); }
    }
    if (slot_ok(eslot)) {
      if (!e_exists) atomicOr(&m.slot_flags[eslot], F_ESDF_PENDING);   // joins the ESDF layer when the EDT of this update runs
      if (!early) first = atomicExch(&m.slot_stamp[eslot], a.mark_pass) != a.mark_pass;
    }
  }
  // TSDF columns of the band: this lane's (x, y) column of block bzz is voxels 64*vx + 8*vy + 0..7 = 64 contiguous bytes
  // (weight 0 -- also what a slot without a TSDF block reads -- contributes nothing)
  int observed = 0, inside = 0, site = 0;
  // one band block's column: 8 voxels {distance, weight} of this lane's (x, y), block z index bzz
  auto column = [&](int32_t bzz, const float4& v0, const float4& v1, const float4& v2, const float4& v3) {
    const float dz[8] = {v0.x, v0.z, v1.x, v1.z, v2.x, v2.z, v3.x, v3.z}, wz[8] = {v0.y, v0.w, v1.y, v1.w, v2.y, v2.w, v3.y, v3.w};
#pragma unroll
    for (int z = 0; z < 8; z++) {
      const int32_t kz = bzz * 8 + z;
      if (kz < kz_lo || kz > kz_hi) continue;
      if (a.site_rule == 2) {          // occupancy layer {log_odds, -}: [U] OccupancySiteFunctor -- known iff log-odds != 0, site = inside = occupied (p > 0.5)
        if (dz[z] != 0.0f) observed = 1;
        if (dz[z] > 0.0f) { inside = 1; site = 1; }
      } else if (wz[z] >= a.min_weight) {
        observed = 1;
        const int in = dz[z] <= 0.0f;
        if (in) inside = 1;
        if ((a.site_rule == 1 || in) && fabsf(dz[z]) <= a.site_dist_m) site = 1;
      }
    }
  };
  if (slot_ok(ts0)) column(bz_lo, p00, p01, p02, p03);                          // (uniform)
  if (nz > 1 && slot_ok(ts1)) column(bz_lo + 1, p10, p11, p12, p13);
  for (int32_t q = 2; q < nz; ++q) {                                            // (a band of more than two blocks: loaded here)
    const uint32_t ts = __shfl(qslot, q + 1);
    if (!slot_ok(ts)) continue;                           // uniform
    const float4* col = reinterpret_cast<const float4*>(&m.tsdf[(size_t)ts * 512 + 64 * vx + 8 * vy]);
    column(bz_lo + q, col[0], col[1], col[2], col[3]);
  }
  eslot = __shfl(eslot, 0); first = __shfl(first, 0);
  NVBX_TV(0, 5, wall_clock64());
  if (!first || !slot_ok(eslot)) return;                // column already re-marked in this marking pass
  if (lane == 0) {                                         // window record: this workgroup's shard copy
    atomicMin(shc_at(m, srec, sh, 0), bx); atomicMin(shc_at(m, srec, sh, 1), by);
    atomicMax(shc_at(m, srec, sh, 2), bx); atomicMax(shc_at(m, srec, sh, 3), by);
    atomicAdd(shc_at(m, srec, sh, 4), 1);
  }
  // the column's masks (bit x + 8y of the block's slice plane); the voxels themselves are written by the EDT only, so a
  // marking pass changes nothing the API can observe
  const u64 sbits = __ballot(site != 0), obits = __ballot(observed != 0), ibits = __ballot(inside != 0);
  if (lane == 0) { m.site_bits[eslot] = sbits; m.obs_bits[eslot] = obits; m.inside_bits[eslot] = ibits; }
}

// `wg` of `nwg` single-wavefront workers; called by k_esdf_mark and by the marking workgroups fused into k_integrate_color.
// Worker w serves shard (w & 7) of the dirty list, entries (w >> 3), (w >> 3) + nwg / 8, ...: the entry's address does not
// depend on the other shards' counts, so the shard's count and the worker's first entry are fetched together (one dependent
// round trip less than a prefix over all eight counts followed by the entry).
__device__ inline void esdf_mark_worker(const DMap& m, const EsdfArgs& a, int wg, int nwg) {
  const int per = nwg >> 3;                               // workers per shard
  const int sh_l = wg & (NSH - 1), j0 = wg >> 3;
  if (per < 1 || j0 >= per) return;
  const int32_t* base = &m.lists[((size_t)S_LIST_ESDF_DIRTY * NSH + sh_l) * m.capacity];
  const int32_t first = base[j0];                         // speculative: valid iff j0 < cnt (j0 < per <= capacity)
  int32_t cnt = *shc_at(m, S_LIST_ESDF_DIRTY, sh_l, 0);
  if (cnt > (int32_t)m.capacity) cnt = (int32_t)m.capacity;
  const int srec = S_ESDF_REC + (int)(a.epoch & 1), sh = my_shard();
  int n_done = 0;
  for (int32_t j = j0; j < cnt; j += per) { esdf_mark_entry(m, a, (uint32_t)(j == j0 ? first : base[j]), srec, sh, true); n_done++;  if (n_done == 1) NVBX_TV(0, 1, wall_clock64()); }
  NVBX_TV(0, 6, n_done); NVBX_TV(0, 2, wall_clock64());
}

// A marking pass that empties the list it consumed (EsdfArgs::self_reset): every one of its `n_workers` wavefronts calls this when it is
// done with its entries; the last one to arrive resets the shard counts and the arrival counter.  Nothing appends to the list while a
// marking pass runs (stream order), so the reset cannot lose an entry.
__device__ inline void esdf_mark_pass_done(const DMap& m, const EsdfArgs& a, int n_workers, int worker = -1) {
  if (!a.self_reset) return;
  // (no fence: the worker's loads from the list have RETURNED -- it used their values -- before it gets here, and the reset is ordered behind
  //  every worker's arrival by the atomics themselves; a __threadfence per worker cost the launch ~3 us)
  if ((threadIdx.x & 63) == 0) {
    // two levels: 256 workers on ONE counter serialise at ~12 ns per atomic (3 us inside a 9 us launch, measured); a worker counts itself in
    // its shard's copy (worker w -> shard w & 7), the last of a shard counts the shard, the last shard resets
    const int w = worker >= 0 ? worker : (int)blockIdx.x, sh = w & (NSH - 1);      // (workers = the launch's first workgroups, unless told otherwise)
    const int32_t in_shard = (n_workers - sh + NSH - 1) / NSH;          // workers w' < n_workers with w' & 7 == sh
    const int32_t arrived = atomicAdd(shc_at(m, S_MARK_DONE, sh, 0), 1);
    if (arrived == in_shard - 1) {
      __hip_atomic_store(shc_at(m, S_MARK_DONE, sh, 0), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t shards_with_workers = n_workers < NSH ? n_workers : NSH;
      const int32_t done = atomicAdd(&m.counters[C_MARK_DONE], 1);
      if (done == shards_with_workers - 1) {
#pragma unroll
        for (int s = 0; s < NSH; s++) __hip_atomic_store(shc_at(m, S_LIST_ESDF_DIRTY, s, 0), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&m.counters[C_MARK_DONE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// The peers' gathered dirty lists (multi-GPU union step): g = int32 [world][1 + max_count][3], row 0 of a rank = its count.
struct ImportArgs { const int32_t* g; int32_t world, self_rank; int64_t max_count; int32_t n_wg; };
// worker `w` of imp.n_wg single-wavefront workers over all peers' entries: look the block up, re-mark its column on the spot
__device__ inline void esdf_import_mark_worker(const DMap& m, const EsdfArgs& a, const ImportArgs& imp, int w) {
  const int srec = S_ESDF_REC + (int)(a.epoch & 1), sh = my_shard();
  const int n_peers = imp.self_rank >= 0 && imp.self_rank < imp.world ? imp.world - 1 : imp.world;
  const int per_rank = max(1, imp.n_wg / max(1, n_peers));       // workers are dealt to the peers in turn
  const int peer = w / per_rank, wi = w - peer * per_rank;
  if (peer >= n_peers) return;
  const int32_t r = (n_peers < imp.world && peer >= imp.self_rank) ? peer + 1 : peer;  // skip the own rank
  const int32_t* base = imp.g + (size_t)r * (size_t)(imp.max_count + 1) * 3;
  int64_t n = base[0]; if (n > imp.max_count) n = imp.max_count;
  const int32_t* idx = base + 3;
  for (int64_t i = wi; i < n; i += per_rank) {
    const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], F_TSDF);
    if (!slot_ok(s)) continue;
    esdf_mark_entry(m, a, s, srec, sh);
  }
}

// The union step WITHOUT the marking, for the two-launch pipeline: thread `t` of `nt` over all peers' entries -- every peer block
// that exists locally as a TSDF block becomes ESDF-dirty (flag + dirty list, de-duplicated by the flag like every other append), to be re-marked
// by the NEXT marking pass.  Rides in the TSDF-update launch: nothing there inserts into the hash, and the update's own appends go through the
// same flag, so a block both updated locally and named by a peer is listed once.
__device__ inline void esdf_import_dirty_worker(const DMap& m, const ImportArgs& imp, int64_t t, int64_t nt) {
  for (int32_t r = 0; r < imp.world; r++) {
    if (r == imp.self_rank) continue;
    const int32_t* base = imp.g + (size_t)r * (size_t)(imp.max_count + 1) * 3;
    int64_t n = base[0]; if (n > imp.max_count) n = imp.max_count;
    const int32_t* idx = base + 3;
    for (int64_t i = t; i < n; i += nt) {
      const uint32_t s = find_slot(m, idx[3 * i], idx[3 * i + 1], idx[3 * i + 2], F_TSDF);
      if (!slot_ok(s)) continue;
      const uint32_t old = atomicOr(&m.slot_flags[s], F_DIRTY_ESDF);
      if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)s);
    }
  }
}

}  // namespace nvbx
