// mesh.hip -- Mapper::updateColorMesh on MI355X: marching cubes, count + weld + emit fused in ONE launch.
//
// One 512-thread workgroup (8 wave64) per TSDF block.  The 9^3 corner lattice (own block + the +x/+y/+z neighbour
// blocks, found by 8 hash probes) is staged in LDS once; thread = cube in z + 8y + 64x order.  Triangle offsets come
// from a block-wide exclusive scan (wave shuffles + 8 partials), vertices are welded by construction: every vertex
// lives on one lattice edge (corner, axis), `atomicMin` in LDS records the first triangle of each crossed edge, a
// second scan over the 2187 possible edges numbers the vertices in ascending edge id and records vertex id -> edge id, so
// that every thread emits exactly one vertex.  A single returning 64-bit atomicAdd per block (vertex | triangle cursor of
// the workgroup's shard region) reserves space in the pre-allocated arenas, so there is no count -> host -> alloc ->
// emit round trip ([U] MeshIntegrator is two-pass with a host sync in between).
// Ordering contract (shared with the oracle): vertices ascending edge id ((lx*9+ly)*9+lz)*3+axis; triangles in cube
// order then table order; vertex normal = normal of the first triangle referencing it; colour = nearest colour voxel.
// Call sites served: nvblox_ros/src/lib/layer_publishing.cpp:686-689, nvblox_node.cpp:1611; output contract
// conversions/mesh_conversions.cpp:62-104.
#include <algorithm>
#include <cstring>
#include <vector>
#include "nvbx_mapper.h"

using namespace nvbx;

// marching-cubes triangle tables, one per ambiguity rule (nvbx_mapper_params::mesh_ambiguity_rule; tools/gen_mc_table.py)
__constant__ int8_t MC_TRI_C[3][256][16] = {{
#include "mc_table.inc"
}, {
#include "mc_table_r1.inc"
}, {
#include "mc_table_r2.inc"
}};
__constant__ int8_t MC_EDGE_BASE_C[12][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,0},{0,0,1},{1,0,1},{0,1,1},{0,0,1},{0,0,0},{1,0,0},{1,1,0},{0,1,0}};
__constant__ int8_t MC_EDGE_AXIS_C[12] = {0,1,0,1,0,1,0,1,2,2,2,2};

constexpr int NLAT = 729, NEDGE = 2187, MAXTRI = 2560;

struct MeshArgs {
  float voxel_size, block_size, min_weight;
  int32_t full;          // 1: every TSDF block, 0: dirty list
  int32_t rule, normal_rule;   // [U] open choices: ambiguity rule (table), welded-vertex normal rule
  int32_t dirty_list;    // S_LIST_* id of the list to mesh
  int32_t next_list;     // the other parity's list (reset here; blocks dirtied from now on go there)
  int32_t rec, rec_next; // C_MESH_OUT records (list entries)
  int32_t srec, srec_next; // S_MESH_REC sharded records (blocks meshed, arena cursors)
  int64_t vert_cap, tri_cap;   // per shard region of the arenas
};

// exclusive scan over the 512 threads of the workgroup; returns this thread's offset, *total = sum
__device__ inline int block_scan_512(int v, int* s_part, int tid, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  __syncthreads();
  if (lane == 63) s_part[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) { const int p = s_part[w]; if (w < wave) base += p; tot += p; }
  *total = tot;
  return base + inc - v;
}

__device__ inline void edge_decode(int eid, int* li, int* lj, int* axis, int* lx, int* ly, int* lz) {
  *axis = eid % 3; *li = eid / 3;
  *lz = *li % 9; *ly = (*li / 9) % 9; *lx = *li / 81;
  *lj = *li + (*axis == 0 ? 81 : (*axis == 1 ? 9 : 1));
}

// Full-layer meshing (UpdateFullLayer::kYes) of a large map is mostly free space: a block whose own and +x/+y/+z neighbour blocks hold
// no observed voxel with a negative distance cannot contain a zero crossing.  A streaming pre-pass (every TSDF voxel read once, eight
// blocks per workgroup iteration, no barriers) leaves one byte per slot -- "has an observed negative voxel" -- and k_mesh then skips
// such blocks after its eight hash probes, before touching a voxel.  (The exact per-slot band flag F_BAND would do the same for
// camera-built maps, but LiDAR integration leaves it stale; the pre-pass is exact for every map and costs ~0.2 ms per 10^5 blocks.)
__global__ __launch_bounds__(512) void k_mesh_prepass(DMap m, float min_weight, uint8_t* neg_any) {
  // 16 bytes (two voxels) per lane and load: a 512-thread workgroup covers two blocks per load instruction, PB of them in flight.
  // No flag test: the TSDF pool of a slot without F_TSDF is all-zero -- weight 0 reads as "unobserved" -- and a flag load in front of
  // every voxel load made each iteration two dependent round trips.
  constexpr int PB = 8;
  const int32_t hw = m.counters[C_HIGH_WATER];
  const int tid = threadIdx.x, half = tid >> 8, q = tid & 255;
  const float4* pool = reinterpret_cast<const float4*>(m.tsdf);
  for (int32_t base = blockIdx.x * (2 * PB); base < hw; base += gridDim.x * (2 * PB)) {
    float4 tv[PB];
#pragma unroll
    for (int j = 0; j < PB; j++) {
      const int32_t slot = base + 2 * j + half;
      tv[j] = slot < hw ? pool[(size_t)slot * 256 + q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int j = 0; j < PB; j++) {
      const bool neg = (tv[j].y >= min_weight && tv[j].x < 0.0f) || (tv[j].w >= min_weight && tv[j].z < 0.0f);
      if (__ballot(neg) != 0ull && (tid & 63) == 0) neg_any[base + 2 * j + half] = 1;       // (only reached with slot < hw: zeros are never negative)
    }
  }
}

#ifndef NVBX_MESH_WAVES
#define NVBX_MESH_WAVES 4      // (waves per SIMD the register budget is sized for: tools/mesh_occupancy_sweep.sh)
#endif
__global__ __launch_bounds__(512, NVBX_MESH_WAVES) void k_mesh(DMap m, MeshArgs a, float* o_vert, float* o_nrm, uint32_t* o_col,
                                              int32_t* o_tri, MeshRecord* o_rec, const uint8_t* neg_any) {
  __shared__ float s_d[NLAT];
  __shared__ uint8_t s_valid[NLAT];
  __shared__ int32_t s_first[NEDGE];
  __shared__ int32_t s_vid[NEDGE];
  __shared__ uint16_t s_tri_edges[MAXTRI * 3];
  __shared__ uint32_t s_nslot[8];
  __shared__ uint16_t s_vedge[NEDGE];     // welded vertex id -> lattice edge id
  __shared__ int s_part[8];
  __shared__ int s_base[3];
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  if (blockIdx.x == 0 && tid < NSH) {                                   // next update's records, one shard per thread
    *shc_at(m, a.next_list, tid, 0) = 0;
    *shc_at(m, a.srec_next, tid, 0) = 0; *shc_at(m, a.srec_next, tid, 2) = 0; *shc_at(m, a.srec_next, tid, 3) = 0;
    if (tid == 0) m.counters[a.rec_next + 0] = 0;
  }
  ListView lv;
  const int32_t nlist = list_open(m, a.dirty_list, &lv);
  const int32_t n = a.full ? m.counters[C_HIGH_WATER] : nlist;
  if (blockIdx.x == 0 && tid == 0) m.counters[a.rec + 0] = n;           // records o_rec[0..n): one per list entry, invalid ones marked
  const int sh = my_shard();
  // one block: record index `it`, pool slot `slot` (whole workgroup; `return` = next block)
  auto mesh_block = [&](const int32_t it, const uint32_t slot) {
    const uint32_t flags = m.slot_flags[slot];
    if (!(flags & F_TSDF)) {                                             // uniform: block was deallocated meanwhile (or, full mode, is no TSDF block)
      if (tid == 0) {
        if (flags & F_DIRTY_MESH) atomicAnd(&m.slot_flags[slot], ~F_DIRTY_MESH);
        MeshRecord r; r.x = INT32_MIN; r.y = 0; r.z = 0; r.vbase = -1; r.nvert = 0; r.tbase = 0; r.ntri = 0; r.pad = 0;
        o_rec[it] = r;
      }
      return;
    }
    const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
    __syncthreads();                                                     // previous iteration done with LDS
    if (tid < 8)      // neighbour blocks: no layer-flag load (a slot without TSDF / colour reads zero weights, nvbx_internal.h)
      s_nslot[tid] = tid == 0 ? slot
                   : (a.min_weight > 0.0f ? any_slot(m, bx + (tid & 1), by + ((tid >> 1) & 1), bz + ((tid >> 2) & 1))
                                          : find_slot(m, bx + (tid & 1), by + ((tid >> 1) & 1), bz + ((tid >> 2) & 1), F_TSDF));   // (weight 0 would pass a min_weight of 0)
    if (tid == 0) { atomicAnd(&m.slot_flags[slot], ~F_DIRTY_MESH); atomicOr(&m.slot_flags[slot], F_MESH); }
    __syncthreads();
    for (int li = tid; li < NLAT; li += 512) {
      const int lz = li % 9, ly = (li / 9) % 9, lx = li / 81;
      const uint32_t ns = s_nslot[(lx >> 3) | ((ly >> 3) << 1) | ((lz >> 3) << 2)];
      float d = 0.0f; uint8_t ok = 0;
      if (slot_ok(ns)) { const float2 tv = m.tsdf[(size_t)ns * 512 + (lz & 7) + 8 * (ly & 7) + 64 * (lx & 7)]; d = tv.x; ok = tv.y >= a.min_weight ? 1 : 0; }
      s_d[li] = d; s_valid[li] = ok;
    }
    __syncthreads();
    // cube classification
    int cube = 0; bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int cx = (c == 1 || c == 2 || c == 5 || c == 6) ? 1 : 0, cy = (c == 2 || c == 3 || c == 6 || c == 7) ? 1 : 0, cz = c >> 2;
      const int li = ((vx + cx) * 9 + (vy + cy)) * 9 + (vz + cz);
      ok = ok && s_valid[li];
      if (s_d[li] < 0.0f) cube |= 1 << c;
    }
    int ntri = 0;
    const int8_t* tri_row = MC_TRI_C[a.rule][cube];
    if (ok && cube != 0 && cube != 255) { while (ntri < 5 && tri_row[3 * ntri] >= 0) ntri++; }
    // a block without a triangle (free space, unobserved): an empty record, no scans, no arena reservation (uniform)
    if (!__syncthreads_or(ntri)) {
      if (tid == 0) {
        atomicAdd(shc_at(m, a.srec, sh, 0), 1);
        MeshRecord r; r.x = bx; r.y = by; r.z = bz; r.vbase = 0; r.nvert = 0; r.tbase = 0; r.ntri = 0; r.pad = 0;
        o_rec[it] = r;
      }
      return;
    }
    for (int e = tid; e < NEDGE; e += 512) s_first[e] = INT32_MAX;
    int T;
    const int toff = block_scan_512(ntri, s_part, tid, &T);      // (its barriers also order the initialisation above before the atomicMin below)
    for (int j = 0; j < ntri; j++) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int e = tri_row[3 * j + q];
        const int eid = (((vx + MC_EDGE_BASE_C[e][0]) * 9 + (vy + MC_EDGE_BASE_C[e][1])) * 9 + (vz + MC_EDGE_BASE_C[e][2])) * 3 + MC_EDGE_AXIS_C[e];
        s_tri_edges[3 * (toff + j) + q] = (uint16_t)eid;
        atomicMin(&s_first[eid], toff + j);
      }
    }
    __syncthreads();
    // vertex numbering: 5 consecutive edge ids per thread (512 * 5 >= 2187), ascending
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { const int e = tid * 5 + k; if (e < NEDGE && s_first[e] != INT32_MAX) cnt++; }
    int V;
    int voff = block_scan_512(cnt, s_part, tid, &V);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int e = tid * 5 + k;
      if (e < NEDGE) { const bool on = s_first[e] != INT32_MAX; if (on) s_vedge[voff] = (uint16_t)e; s_vid[e] = on ? voff++ : -1; }
    }
    if (tid == 0) {
      // ONE returning atomic per block: the vertex and triangle arena cursors share a 64-bit word (three separate
      // counters cost 3 x ~12 ns x #blocks of serialised L2 atomics: 8.4 of the kernel's 28 us at 300 blocks)
      // The arenas are split into NSH regions with one cursor each (this workgroup's shard), so concurrent blocks
      // do not serialise on one word.
      const u64 cur = atomicAdd(reinterpret_cast<u64*>(shc_at(m, a.srec, sh, 2)), (u64)(uint32_t)V | ((u64)(uint32_t)T << 32));
      const int64_t vl = (int64_t)(cur & 0xFFFFFFFFull), tl = (int64_t)(cur >> 32);
      atomicAdd(shc_at(m, a.srec, sh, 0), 1);                             // blocks meshed (not waited for)
      const int bi = it;
      int nv = V, nt = T;
      int vb = (int)((int64_t)sh * a.vert_cap + vl), tb = (int)((int64_t)sh * a.tri_cap + tl);
      if (vl + V > a.vert_cap || tl + T > a.tri_cap) { atomicExch(&m.counters[C_OVERFLOW], 1); nv = 0; nt = 0; vb = -1; }
      MeshRecord r; r.x = bx; r.y = by; r.z = bz; r.vbase = vb; r.nvert = nv; r.tbase = tb; r.ntri = nt; r.pad = 0;
      o_rec[bi] = r;
      s_base[0] = vb; s_base[1] = tb;
    }
    __syncthreads();
    const int vbase = s_base[0], tbase = s_base[1];
    if (vbase < 0) return;                                             // uniform (arena overflow)
    // emit vertices
    const int32_t b3[3] = {bx, by, bz};
    // one welded vertex per thread (a block has ~25-150 of them): balanced, unlike walking the 2187 edge ids
#pragma unroll 1
    for (int vid = tid; vid < V; vid += 512) {
      const int e = s_vedge[vid];
      int li, lj, axis, lx, ly, lz;
      edge_decode(e, &li, &lj, &axis, &lx, &ly, &lz);
      const float da = s_d[li], db = s_d[lj];
      const float t = da / (da - db);
      const int32_t l3[3] = {lx, ly, lz};
      float p[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        float pos = ((float)b3[q] * a.block_size + (float)l3[q] * a.voxel_size) + a.voxel_size * 0.5f;
        if (q == axis) pos = pos + t * a.voxel_size;
        p[q] = pos;
      }
      // colour: nearest lattice end point
      const int lc = (t < 0.5f) ? li : lj;
      const int cz = lc % 9, cy = (lc / 9) % 9, cx = lc / 81;
      const int nb = (cx >> 3) | ((cy >> 3) << 1) | ((cz >> 3) << 2);
      uint32_t rgba = 127u | (127u << 8) | (127u << 16);
      if (slot_ok(s_nslot[nb])) {
        const uint2 cv = m.color[(size_t)s_nslot[nb] * 512 + (cz & 7) + 8 * (cy & 7) + 64 * (cx & 7)];
        if (__uint_as_float(cv.y) > 0.0f) rgba = cv.x & 0x00FFFFFFu;
      }
      rgba |= 255u << 24;
      // normal: rule 0 = the first triangle referencing this vertex; rule 1 = area-weighted mean (sum of the unnormalised
      // triangle normals, ascending triangle index) of the block's triangles referencing it
      float nn[3] = {0.0f, 0.0f, 0.0f};
      const int t_lo = a.normal_rule == 0 ? s_first[e] : 0, t_hi = a.normal_rule == 0 ? s_first[e] + 1 : T;
#pragma unroll 1
      for (int tf = t_lo; tf < t_hi; tf++) {
        if (a.normal_rule != 0 && s_tri_edges[3 * tf] != e && s_tri_edges[3 * tf + 1] != e && s_tri_edges[3 * tf + 2] != e) continue;
        float tp[3][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int eq = s_tri_edges[3 * tf + q];
          int qi, qj, qa, qx, qy, qz;
          edge_decode(eq, &qi, &qj, &qa, &qx, &qy, &qz);
          const float ea = s_d[qi], eb = s_d[qj];
          const float tt = ea / (ea - eb);
          const int32_t q3[3] = {qx, qy, qz};
#pragma unroll
          for (int w = 0; w < 3; w++) {
            float pos = ((float)b3[w] * a.block_size + (float)q3[w] * a.voxel_size) + a.voxel_size * 0.5f;
            if (w == qa) pos = pos + tt * a.voxel_size;
            tp[q][w] = pos;
          }
        }
        const float e1[3] = {tp[1][0] - tp[0][0], tp[1][1] - tp[0][1], tp[1][2] - tp[0][2]};
        const float e2[3] = {tp[2][0] - tp[0][0], tp[2][1] - tp[0][1], tp[2][2] - tp[0][2]};
        nn[0] = nn[0] + (e1[1] * e2[2] - e1[2] * e2[1]); nn[1] = nn[1] + (e1[2] * e2[0] - e1[0] * e2[2]); nn[2] = nn[2] + (e1[0] * e2[1] - e1[1] * e2[0]);
      }
      const float len = sqrtf((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]);
      if (len > 0.0f) { nn[0] = nn[0] / len; nn[1] = nn[1] / len; nn[2] = nn[2] / len; }
      const size_t o = (size_t)(vbase + vid);
      o_vert[3 * o] = p[0]; o_vert[3 * o + 1] = p[1]; o_vert[3 * o + 2] = p[2];
      o_nrm[3 * o] = nn[0]; o_nrm[3 * o + 1] = nn[1]; o_nrm[3 * o + 2] = nn[2];
      o_col[o] = rgba;
    }
    // emit triangles (indices local to the block)
    for (int j = 0; j < ntri; j++) {
      const size_t o = (size_t)(tbase + toff + j);
      o_tri[3 * o] = s_vid[s_tri_edges[3 * (toff + j)]];
      o_tri[3 * o + 1] = s_vid[s_tri_edges[3 * (toff + j) + 1]];
      o_tri[3 * o + 2] = s_vid[s_tri_edges[3 * (toff + j) + 2]];
    }
  };
  if (!a.full) {
    for (int32_t it = blockIdx.x; it < n; it += gridDim.x) mesh_block(it, (uint32_t)list_at(m, a.dirty_list, lv, it));
    return;
  }
  // Full layer (record index == slot).  Most of a large map is free space, and per block the test "can it hold a zero crossing" is a
  // chain of dependent loads (flags -> Index3D -> 7 hash probes -> 8 bytes of neg_any): one block per workgroup iteration made the
  // full-layer launch latency-bound.  So the first wavefront tests 64 consecutive slots at once, one per lane, writes the empty
  // records of the blocks without a negative voxel in reach itself, and only the candidates run the meshing path.
  __shared__ u64 s_cand;
  for (int32_t base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    __syncthreads();
    if (tid < 64) {
      const int32_t slot = base + tid;
      bool tsdf = false, reach = false;
      if (slot < n) {
        const uint32_t flags = m.slot_flags[slot];
        tsdf = (flags & F_TSDF) != 0;
        MeshRecord r; r.x = INT32_MIN; r.y = 0; r.z = 0; r.vbase = -1; r.nvert = 0; r.tbase = 0; r.ntri = 0; r.pad = 0;
        if (tsdf) {
          const int32_t bx = m.slot_index[3 * slot], by = m.slot_index[3 * slot + 1], bz = m.slot_index[3 * slot + 2];
          reach = neg_any[slot] != 0;
#pragma unroll
          for (int q = 1; q < 8; q++) {
            const uint32_t ns = any_slot(m, bx + (q & 1), by + ((q >> 1) & 1), bz + ((q >> 2) & 1));
            if (slot_ok(ns) && neg_any[ns]) reach = true;
          }
          if (!reach) {                                    // no zero crossing possible: the empty record of a meshed block
            atomicAnd(&m.slot_flags[slot], ~F_DIRTY_MESH); atomicOr(&m.slot_flags[slot], F_MESH);
            r.x = bx; r.y = by; r.z = bz; r.vbase = 0;
            o_rec[slot] = r;
          }
        } else {
          if (flags & F_DIRTY_MESH) atomicAnd(&m.slot_flags[slot], ~F_DIRTY_MESH);
          o_rec[slot] = r;
        }
      }
      const u64 skipped = __ballot(tsdf && !reach), cand = __ballot(tsdf && reach);
      if (tid == 0) { if (skipped) atomicAdd(shc_at(m, a.srec, sh, 0), (int32_t)__popcll(skipped)); s_cand = cand; }
    }
    __syncthreads();
    u64 cand = s_cand;
    while (cand) {
      const int cj = __ffsll((long long)cand) - 1;
      cand &= cand - 1ull;
      mesh_block(base + cj, (uint32_t)(base + cj));
    }
  }
}

extern "C" int nvbx_update_color_mesh(nvbx_mapper* m, int32_t update_full_layer) {
  if (!m) return NVBX_E_INVALID;
  NVBX_HIP(hipSetDevice(m->device));
  if (m->p.projective_layer_type == 1) return NVBX_OK;      // "Mesh integration is not implemented for occupancy layers" (nvblox_node.cpp:186)
  if (m->join_side()) return NVBX_E_DEVICE;
  MeshArgs a{};
  a.voxel_size = m->p.voxel_size; a.block_size = m->p.voxel_size * 8.0f; a.min_weight = m->p.mesh_min_weight;
  a.full = update_full_layer ? 1 : 0;
  a.rule = m->p.mesh_ambiguity_rule; a.normal_rule = m->p.mesh_normal_rule;
  const int par = (int)(m->mesh_epoch & 1);
  a.dirty_list = S_LIST_MESH_DIRTY + par; a.next_list = S_LIST_MESH_DIRTY + (par ^ 1);
  a.rec = C_MESH_OUT + 4 * par; a.rec_next = C_MESH_OUT + 4 * (par ^ 1);
  a.srec = S_MESH_REC + par; a.srec_next = S_MESH_REC + (par ^ 1);
  a.vert_cap = m->mesh_vert_cap / NSH; a.tri_cap = m->mesh_tri_cap / NSH;
  // 37 KB of LDS per workgroup -> 4 resident per CU: keep the grid within one resident batch (a 2048-workgroup grid ran
  // as two batches, the second waiting ~10 us for the first to drain); longer lists are covered by the grid-stride loop
  const int grid = (int)std::min<int64_t>(m->capacity, 768);
  const uint8_t* neg_any = nullptr;
  if (a.full) {            // one byte per slot in the export scratch (capacity x 12 bytes)
    NVBX_HIP(hipMemsetAsync(m->export_idx, 0, (size_t)m->capacity, m->stream));
    NVBX_LAUNCH(m, k_mesh_prepass, dim3((unsigned)std::min<int64_t>((m->capacity + 15) / 16, 2048)), dim3(512), m->d, a.min_weight, (uint8_t*)m->export_idx);
    neg_any = (const uint8_t*)m->export_idx;
  }
  NVBX_LAUNCH(m, k_mesh, dim3(grid), dim3(512), m->d, a, m->mesh_vert, m->mesh_nrm, (uint32_t*)m->mesh_col,
                     m->mesh_tri, m->mesh_rec, neg_any);
  NVBX_HIP(hipGetLastError());
  m->mesh_epoch++;
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ output
extern "C" int nvbx_mesh_sizes(nvbx_mapper* m, int64_t* n_blocks, int64_t* n_vertices, int64_t* n_triangles) {
  if (!m || !n_blocks || !n_vertices || !n_triangles) return NVBX_E_INVALID;
  if (m->mesh_epoch == 0) { *n_blocks = *n_vertices = *n_triangles = 0; return NVBX_OK; }
  if (m->fetch_counters()) return NVBX_E_DEVICE;
  const int par = (int)((m->mesh_epoch + 1) & 1);
  *n_blocks = m->shc_sum(S_MESH_REC + par, 0);
  int64_t nv = 0, nt = 0;
  for (int s = 0; s < NSH; s++) {      // a shard region that overflowed holds at most its capacity
    nv += std::min<int64_t>((uint32_t)m->h_shc[((S_MESH_REC + par) * NSH + s) * SH_STRIDE + 2], m->mesh_vert_cap / NSH);
    nt += std::min<int64_t>((uint32_t)m->h_shc[((S_MESH_REC + par) * NSH + s) * SH_STRIDE + 3], m->mesh_tri_cap / NSH);
  }
  *n_vertices = nv; *n_triangles = nt;
  return NVBX_OK;
}

// Copies the last update's mesh to host memory, re-packed in ascending (x,y,z) block order so the output is
// deterministic although arena placement is not.
extern "C" int nvbx_mesh_copy(nvbx_mapper* m, nvbx_index3d* block_indices, int32_t* vertex_offsets, int32_t* triangle_offsets, float* vertices,
                              float* normals, uint8_t* colors, int32_t* triangles) {
  if (!m) return NVBX_E_INVALID;
  int64_t nb, nv, nt;
  int rc = nvbx_mesh_sizes(m, &nb, &nv, &nt); if (rc) return rc;
  if (nb == 0) { if (vertex_offsets) vertex_offsets[0] = 0; if (triangle_offsets) triangle_offsets[0] = 0; return NVBX_OK; }
  const int64_t nraw = m->h_counters[C_MESH_OUT + 4 * (int)((m->mesh_epoch + 1) & 1) + 0];     // one record per list entry
  std::vector<MeshRecord> rec((size_t)nraw);
  NVBX_HIP(hipMemcpy(rec.data(), m->mesh_rec, (size_t)nraw * sizeof(MeshRecord), hipMemcpyDeviceToHost));
  rec.erase(std::remove_if(rec.begin(), rec.end(), [](const MeshRecord& r) { return r.x == INT32_MIN; }), rec.end());
  if ((int64_t)rec.size() != nb) { set_error("mesh record count mismatch"); return NVBX_E_DEVICE; }
  // download the used prefix of every shard region of the arenas; records address them by absolute position
  const int par = (int)((m->mesh_epoch + 1) & 1);
  const int64_t vreg = m->mesh_vert_cap / NSH, treg = m->mesh_tri_cap / NSH;
  std::vector<int64_t> voff(NSH + 1, 0), toff(NSH + 1, 0);
  for (int s = 0; s < NSH; s++) {
    voff[s + 1] = voff[s] + std::min<int64_t>((uint32_t)m->h_shc[((S_MESH_REC + par) * NSH + s) * SH_STRIDE + 2], vreg);
    toff[s + 1] = toff[s] + std::min<int64_t>((uint32_t)m->h_shc[((S_MESH_REC + par) * NSH + s) * SH_STRIDE + 3], treg);
  }
  std::vector<float> v((size_t)nv * 3 + 3), n((size_t)nv * 3 + 3); std::vector<uint8_t> c((size_t)nv * 4 + 4); std::vector<int32_t> t((size_t)nt * 3 + 3);
  for (int s = 0; s < NSH; s++) {
    const int64_t cv = voff[s + 1] - voff[s], ct = toff[s + 1] - toff[s];
    if (cv) {
      NVBX_HIP(hipMemcpy(v.data() + voff[s] * 3, m->mesh_vert + (size_t)s * vreg * 3, (size_t)cv * 12, hipMemcpyDeviceToHost));
      NVBX_HIP(hipMemcpy(n.data() + voff[s] * 3, m->mesh_nrm + (size_t)s * vreg * 3, (size_t)cv * 12, hipMemcpyDeviceToHost));
      NVBX_HIP(hipMemcpy(c.data() + voff[s] * 4, m->mesh_col + (size_t)s * vreg * 4, (size_t)cv * 4, hipMemcpyDeviceToHost));
    }
    if (ct) NVBX_HIP(hipMemcpy(t.data() + toff[s] * 3, m->mesh_tri + (size_t)s * treg * 3, (size_t)ct * 12, hipMemcpyDeviceToHost));
  }
  // absolute arena position -> position in the packed host copies
  for (auto& r : rec) {
    if (r.vbase < 0) continue;
    const int sv = (int)(r.vbase / vreg), st = (int)(r.tbase / treg);
    r.vbase = (int32_t)(voff[sv] + (r.vbase - (int64_t)sv * vreg));
    r.tbase = (int32_t)(toff[st] + (r.tbase - (int64_t)st * treg));
  }
  std::vector<int> order((size_t)nb);
  for (int64_t i = 0; i < nb; i++) order[(size_t)i] = (int)i;
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    const MeshRecord &p = rec[(size_t)a], &q = rec[(size_t)b];
    if (p.x != q.x) return p.x < q.x; if (p.y != q.y) return p.y < q.y; return p.z < q.z; });
  int64_t vo = 0, to = 0;
  for (int64_t i = 0; i < nb; i++) {
    const MeshRecord& r = rec[(size_t)order[(size_t)i]];
    if (block_indices) { block_indices[i].x = r.x; block_indices[i].y = r.y; block_indices[i].z = r.z; }
    if (vertex_offsets) vertex_offsets[i] = (int32_t)vo;
    if (triangle_offsets) triangle_offsets[i] = (int32_t)to;
    if (r.nvert > 0) {
      if (vertices) memcpy(vertices + vo * 3, v.data() + (size_t)r.vbase * 3, (size_t)r.nvert * 12);
      if (normals) memcpy(normals + vo * 3, n.data() + (size_t)r.vbase * 3, (size_t)r.nvert * 12);
      if (colors) memcpy(colors + vo * 4, c.data() + (size_t)r.vbase * 4, (size_t)r.nvert * 4);
    }
    if (r.ntri > 0 && triangles) memcpy(triangles + to * 3, t.data() + (size_t)r.tbase * 3, (size_t)r.ntri * 12);
    vo += r.nvert; to += r.ntri;
  }
  if (vertex_offsets) vertex_offsets[nb] = (int32_t)vo;
  if (triangle_offsets) triangle_offsets[nb] = (int32_t)to;
  return NVBX_OK;
}
