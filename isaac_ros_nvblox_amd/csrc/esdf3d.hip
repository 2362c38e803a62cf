// esdf3d.hip -- EsdfMode::k3D (node param esdf_mode "3d", node_params.hpp:90; MultiMapper(voxel_size, mapping_type, EsdfMode, ...),
// nvblox_node.cpp:187-190): the ESDF of every voxel of every updated block -- what the EsdfAndGradients query
// (esdf_and_gradients_conversions.cu:88-125) samples for manipulators.  Not the benchmark path (both shipped configurations
// run "2d"), so this is built for exactness and simplicity, not for the last microsecond:
//   k_esdf3_mark   one 512-thread workgroup per dirty block: observed / inside / site of each voxel from its own TSDF voxel (or
//                  log-odds), written into the ESDF voxel's flag bits; the ESDF block is the TSDF block's own slot.
//   window         = AABB of the marked (and dropped) blocks + R; the host reads it (one sync) and sizes a dense scratch volume
//                  of window + R: site bits, then the exact separable Euclidean distance transform with cut-off --
//   k_esdf3_x      nearest site along x per voxel (ties -> -x),
//   k_esdf3_y / z  min over +-ri of d^2 + previous pass, increasing |d| with early exit, ties -> smaller offset (packed-key min),
//                  the z pass writes {squared distance, parent direction} into the ESDF blocks of the window.
// Semantics and tie rules are the oracle's update_esdf_3d (oracle/nvblox_oracle.c); every value compares bit-exactly.
#include <algorithm>
#include <cmath>
#include "nvbx_mapper.h"

using namespace nvbx;

struct Win3 { int32_t bx0, by0, bz0, nbx, nby, nbz; };     // scratch volume in blocks (window + R)
constexpr uint32_t B_NONE = 0xFFFFFFFFu;

__global__ __launch_bounds__(512) void k_esdf3_mark(DMap m, EsdfArgs a) {
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;          // TSDF voxel order z + 8y + 64x
  ListView lv;
  const int32_t n = list_open(m, S_LIST_ESDF_DIRTY, &lv);
  for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t s = (uint32_t)list_at(m, S_LIST_ESDF_DIRTY, lv, i);
    const uint32_t flags = m.slot_flags[s];
    if (!(flags & (F_TSDF | F_ESDF))) continue;                        // freed since it was listed (uniform)
    const float2 tv = m.tsdf[(size_t)s * 512 + tid];                   // all-zero if the TSDF block is gone
    int observed = 0, inside = 0, site = 0;
    if (a.site_rule == 2) { if (tv.x != 0.0f) observed = 1; if (tv.x > 0.0f) { inside = 1; site = 1; } }
    else if ((flags & F_TSDF) && tv.y >= a.min_weight) {
      observed = 1;
      const int in = tv.x <= 0.0f;
      if (in) inside = 1;
      if ((a.site_rule == 1 || in) && fabsf(tv.x) <= a.site_dist_m) site = 1;
    }
    uint2* ev = &m.esdf[(size_t)s * 512 + vx + 8 * vy + 64 * vz];
    const uint2 old = *ev;
    *ev = make_uint2(old.x, (old.y & ~ESDF_FLAG_MASK) | (observed ? ESDF_OBSERVED : 0u) | (inside ? ESDF_INSIDE : 0u) | (site ? ESDF_SITE : 0u));
    if (tid == 0) {
      atomicOr(&m.slot_flags[s], F_ESDF);
      atomicAnd(&m.slot_flags[s], ~(F_DIRTY_ESDF | F_ESDF_REMARK));
      const int32_t bx = m.slot_index[3 * s], by = m.slot_index[3 * s + 1], bz = m.slot_index[3 * s + 2];
      if (bz == a.bz_out) {        // the slicer's image covers the blocks of the slice plane
        atomicMin(&m.counters[C_ESDF_AABB + 0], bx); atomicMin(&m.counters[C_ESDF_AABB + 1], by);
        atomicMax(&m.counters[C_ESDF_AABB + 2], bx); atomicMax(&m.counters[C_ESDF_AABB + 3], by);
      }
      atomicMin(&m.counters[C_ESDF3_WIN + 0], bx); atomicMin(&m.counters[C_ESDF3_WIN + 1], by); atomicMin(&m.counters[C_ESDF3_WIN + 2], bz);
      atomicMax(&m.counters[C_ESDF3_WIN + 3], bx); atomicMax(&m.counters[C_ESDF3_WIN + 4], by); atomicMax(&m.counters[C_ESDF3_WIN + 5], bz);
      atomicAdd(&m.counters[C_ESDF3_WIN + 6], 1);
    }
  }
}
__global__ void k_esdf3_reset(DMap m) {
  const int t = threadIdx.x;
  if (t < 3) m.counters[C_ESDF3_WIN + t] = INT32_MAX; else if (t < 6) m.counters[C_ESDF3_WIN + t] = INT32_MIN; else if (t == 6) m.counters[C_ESDF3_WIN + 6] = 0;
  if (t < NSH) *shc_at(m, S_LIST_ESDF_DIRTY, t, 0) = 0;
}

// site bits of the scratch volume: one byte per (block, y, z) row of 8 voxels along x.  One wavefront per block, lane = row.
__global__ __launch_bounds__(64) void k_esdf3_bits(DMap m, Win3 w, uint8_t* bits) {
  const int lane = threadIdx.x, y = lane & 7, z = lane >> 3;
  const int64_t nblk = (int64_t)w.nbx * w.nby * w.nbz;
  const int64_t H = (int64_t)w.nby * 8;
  for (int64_t c = blockIdx.x; c < nblk; c += gridDim.x) {
    const int32_t cx = (int32_t)(c % w.nbx), cy = (int32_t)((c / w.nbx) % w.nby), cz = (int32_t)(c / ((int64_t)w.nbx * w.nby));
    const uint32_t s = find_slot(m, w.bx0 + cx, w.by0 + cy, w.bz0 + cz, F_ESDF);
    uint32_t byte = 0;
    if (slot_ok(s)) {
      const uint2* row = &m.esdf[(size_t)s * 512 + 8 * y + 64 * z];    // x = 0..7 contiguous
#pragma unroll
      for (int x = 0; x < 8; x++) if (row[x].y & ESDF_SITE) byte |= 1u << x;
    }
    bits[(((int64_t)cz * 8 + z) * H + (int64_t)cy * 8 + y) * w.nbx + cx] = (uint8_t)byte;
  }
}
__device__ inline bool bit_at(const uint8_t* row_bits, int32_t nbx, int64_t x) {       // x in voxels along the row
  if (x < 0 || x >= (int64_t)nbx * 8) return false;
  return (row_bits[x >> 3] >> (x & 7)) & 1u;
}
// x pass: one thread per (row, block): offset to the nearest site along x for its 8 voxels (127 = none within ri; ties -> -x)
__global__ __launch_bounds__(256) void k_esdf3_x(Win3 w, const uint8_t* bits, int8_t* A, int32_t ri) {
  const int64_t rows = (int64_t)w.nbz * 8 * w.nby * 8;
  const int64_t n = rows * w.nbx;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / w.nbx; const int32_t cx = (int32_t)(i - r * w.nbx);
    const uint8_t* rb = bits + r * w.nbx;
    int8_t out[8];
#pragma unroll
    for (int x = 0; x < 8; x++) {
      const int64_t p = (int64_t)cx * 8 + x;
      int best = 127;
      for (int32_t d = 0; d <= ri; d++) {
        if (bit_at(rb, w.nbx, p - d)) { best = -d; break; }
        if (bit_at(rb, w.nbx, p + d)) { best = d; break; }
      }
      out[x] = (int8_t)best;
    }
    *reinterpret_cast<uint2*>(A + r * ((int64_t)w.nbx * 8) + (int64_t)cx * 8) = *reinterpret_cast<const uint2*>(out);
  }
}
// y pass: one thread per voxel: min over dy of dy^2 + dx(y + dy)^2, key = sq << 16 | (dy + 128) << 8 | (dx + 128)
__global__ __launch_bounds__(256) void k_esdf3_y(Win3 w, const int8_t* A, uint32_t* B, int32_t ri) {
  const int64_t W = (int64_t)w.nbx * 8, H = (int64_t)w.nby * 8, D = (int64_t)w.nbz * 8;
  const int64_t n = W * H * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = i % W, y = (i / W) % H, z = i / (W * H);
    const int8_t* col = A + z * H * W + x;
    uint32_t best = B_NONE;
    for (int32_t k = 0; k <= ri; k++) {
      if (best != B_NONE && (uint32_t)(k * k) > (best >> 16)) break;
      for (int sgn = (k == 0 ? 1 : -1); sgn <= 1; sgn += 2) {
        const int64_t yy = y + (int64_t)sgn * k;
        if (yy < 0 || yy >= H) continue;
        const int dx = col[yy * W];
        if (dx == 127) continue;
        const uint32_t key = ((uint32_t)(k * k + dx * dx) << 16) | ((uint32_t)(sgn * k + 128) << 8) | (uint32_t)(dx + 128);
        if (key < best) best = key;
      }
    }
    B[i] = best;
  }
}
// z pass: one workgroup per block of the WINDOW (scratch minus its R margin) that carries an ESDF block; thread = voxel
__global__ __launch_bounds__(512) void k_esdf3_z(DMap m, Win3 w, int32_t margin, const uint32_t* B, int32_t ri, float max_sq) {
  const int tid = threadIdx.x;
  const int x = tid & 7, y = (tid >> 3) & 7, z = tid >> 6;              // ESDF voxel order x + 8y + 64z
  const int32_t ox = w.nbx - 2 * margin, oy = w.nby - 2 * margin, oz = w.nbz - 2 * margin;
  const int64_t W = (int64_t)w.nbx * 8, H = (int64_t)w.nby * 8, D = (int64_t)w.nbz * 8;
  const int64_t nblk = (int64_t)ox * oy * oz;
  for (int64_t c = blockIdx.x; c < nblk; c += gridDim.x) {
    const int32_t cx = (int32_t)(c % ox) + margin, cy = (int32_t)((c / ox) % oy) + margin, cz = (int32_t)(c / ((int64_t)ox * oy)) + margin;
    const uint32_t s = find_slot(m, w.bx0 + cx, w.by0 + cy, w.bz0 + cz, F_ESDF);
    if (!slot_ok(s)) continue;                                          // uniform
    const int64_t gx = (int64_t)cx * 8 + x, gy = (int64_t)cy * 8 + y, gz = (int64_t)cz * 8 + z;
    const uint32_t* colz = B + gy * W + gx;
    uint64_t best = ~0ull;                                              // sq << 32 | (dz + 128) << 24 | (dy + 128) << 8 | (dx + 128)
    for (int32_t k = 0; k <= ri; k++) {
      if (best != ~0ull && (uint64_t)(k * k) > (best >> 32)) break;
      for (int sgn = (k == 0 ? 1 : -1); sgn <= 1; sgn += 2) {
        const int64_t zz = gz + (int64_t)sgn * k;
        if (zz < 0 || zz >= D) continue;
        const uint32_t b = colz[zz * H * W];
        if (b == B_NONE) continue;
        const uint64_t key = ((uint64_t)((uint32_t)(k * k) + (b >> 16)) << 32) | ((uint64_t)(uint32_t)(sgn * k + 128) << 24) | (uint64_t)(b & 0xFFFFu);
        if (key < best) best = key;
      }
    }
    uint2* ev = &m.esdf[(size_t)s * 512 + tid];
    const uint32_t fl = ev->y & ESDF_FLAG_MASK;
    float sq = max_sq; int dx = 0, dy = 0, dz = 0;
    if (best != ~0ull && (float)(uint32_t)(best >> 32) <= max_sq) {
      sq = (float)(uint32_t)(best >> 32);
      dz = (int)((best >> 24) & 0xFF) - 128; dy = (int)((best >> 8) & 0xFF) - 128; dx = (int)(best & 0xFF) - 128;
    }
    *ev = make_uint2(__float_as_uint(sq), (esdf_meta(dx, dy, dz, 0, 0, 0) & ~ESDF_FLAG_MASK) | fl);
  }
}

int nvbx_mapper::update_esdf_3d() {
  const EsdfArgs a = make_esdf_args();
  if (dirty_since_mark) NVBX_LAUNCH(this, k_esdf3_mark, dim3((unsigned)std::min<int64_t>(capacity, 1024)), dim3(512), d, a);
  dirty_since_mark = false; premark_consumed = false;
  if (fetch_counters()) return NVBX_E_DEVICE;                    // the window is sized on the host: one synchronisation per 3-D update
  const int32_t* c = h_counters + C_ESDF3_WIN;
  esdf3_blocks_marked = c[6];
  if (c[0] > c[3]) { NVBX_LAUNCH(this, k_esdf3_reset, dim3(1), dim3(64), d); esdf_epoch++; return NVBX_OK; }
  const int32_t rb = a.rb;
  Win3 w;                                                        // scratch = changed blocks + 2R (the window + R is recomputed, + R more feeds it)
  w.bx0 = c[0] - 2 * rb; w.by0 = c[1] - 2 * rb; w.bz0 = c[2] - 2 * rb;
  w.nbx = c[3] - c[0] + 1 + 4 * rb; w.nby = c[4] - c[1] + 1 + 4 * rb; w.nbz = c[5] - c[2] + 1 + 4 * rb;
  const int64_t nblk = (int64_t)w.nbx * w.nby * w.nbz, nvox = nblk * 512;
  if (nvox > (1ll << 31)) { set_error("3-D ESDF update: the changed region (+ 2 R) exceeds 2^31 voxels; update more often or lower esdf_max_distance_m"); return NVBX_E_CAPACITY; }
  const int64_t need = nvox / 8 + nvox + nvox * 4 + 1024;
  if (need > esdf3_scratch_bytes) {
    if (esdf3_scratch) NVBX_HIP(hipFree(esdf3_scratch));
    esdf3_scratch = nullptr; esdf3_scratch_bytes = 0;
    NVBX_HIP(hipMalloc(&esdf3_scratch, (size_t)need));
    esdf3_scratch_bytes = need;
  }
  uint32_t* B = (uint32_t*)esdf3_scratch;                         // 4 B / voxel first (alignment), then 1 B / voxel, then the bits
  int8_t* A = (int8_t*)esdf3_scratch + nvox * 4;
  uint8_t* bits = (uint8_t*)esdf3_scratch + nvox * 5;
  const unsigned gblk = (unsigned)std::min<int64_t>(nblk, 1 << 16);
  NVBX_LAUNCH(this, k_esdf3_bits, dim3(gblk), dim3(64), d, w, bits);
  NVBX_LAUNCH(this, k_esdf3_x, dim3((unsigned)std::min<int64_t>((nvox / 8 + 255) / 256, 1 << 16)), dim3(256), w, (const uint8_t*)bits, A, a.ri);
  NVBX_LAUNCH(this, k_esdf3_y, dim3((unsigned)std::min<int64_t>((nvox + 255) / 256, 1 << 18)), dim3(256), w, (const int8_t*)A, B, a.ri);
  NVBX_LAUNCH(this, k_esdf3_z, dim3(gblk), dim3(512), d, w, rb, (const uint32_t*)B, a.ri, a.max_sq);
  NVBX_LAUNCH(this, k_esdf3_reset, dim3(1), dim3(64), d);
  NVBX_HIP(hipGetLastError());
  esdf3_window_voxels = (int64_t)(w.nbx - 2 * rb) * (w.nby - 2 * rb) * (w.nbz - 2 * rb) * 512;
  esdf_epoch++;
  return NVBX_OK;
}
