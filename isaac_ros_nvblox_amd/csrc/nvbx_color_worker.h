// nvbx_color_worker.h -- the colour-integration worker of MultiMapper::integrateColor (ProjectiveColorIntegrator::integrateFrame restated),
// shared by k_integrate_color (color.hip) and by the fused colour + TSDF launch of the pipelined order (tsdf.hip).
#pragma once
#include "nvbx_mapper.h"
#include "nvbx_sphere_trace.h"

namespace nvbx {

template <typename Pix, int NB> struct FrameSetC { FrameCore f[NB]; Pix img[NB]; int32_t n; int32_t chunk; };

// colour source: rgb8 (nvblox::Color, 3 bytes) or bgra8 (4 bytes, channel reorder of ToRgba<Bgra> fused into the fetch)
struct PixRgb8 {
  const uint8_t* p;
  __device__ void tap(int32_t i, float* c) const { const uint8_t* q = p + (int64_t)i * 3; c[0] = (float)q[0]; c[1] = (float)q[1]; c[2] = (float)q[2]; }
};
struct PixBgra8 {
  const uint32_t* p;     // little endian: b | g << 8 | r << 16 | a << 24
  __device__ void tap(int32_t i, float* c) const { const uint32_t v = p[i]; c[0] = (float)((v >> 16) & 0xFF); c[1] = (float)((v >> 8) & 0xFF); c[2] = (float)(v & 0xFF); }
};

__device__ inline uint32_t blend_u8(float c0, float w0, float c1, float w1) {
  const float tw = w0 + w1;
  const float a = NVBX_DIV(w0, tw), b = NVBX_DIV(w1, tw);
  float v = c0 * a + c1 * b;
  v = floorf(v + 0.5f);
  if (v < 0.0f) v = 0.0f;
  if (v > 255.0f) v = 255.0f;
  return (uint32_t)v;
}

// one block of the colour frame(s): `in_view` = the cameras whose frustum the block touches (a non-empty, workgroup-uniform mask); every
// thread of the 512-thread workgroup calls (lane = voxel)
template <typename Pix, int NB>
__device__ inline void color_integrate_block(const DMap& m, const FrameSetC<Pix, NB>& fs, const float* synth_all, int32_t srows, int32_t scols, int32_t mesh_list,
                                             const int32_t slot, const int32_t bx, const int32_t by, const int32_t bz, const uint32_t in_view) {
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  const FrameCore& f0 = fs.f[0];
  const int ncam = NB > 1 ? fs.n : 1;
  if (tid == 0) {
    const uint32_t old = atomicOr(&m.slot_flags[slot], F_COLOR | F_DIRTY_MESH);
    if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, slot);
    if ((in_view >> (ncam - 1)) & 1u) list_append(m, S_LIST_COLOR, slot);    // "the last colour view" = the last camera's, as separate calls would leave it
  }
  const float lx = voxel_center(bx, vx, f0.block_size, f0.voxel_size), ly = voxel_center(by, vy, f0.block_size, f0.voxel_size),
              lz = voxel_center(bz, vz, f0.block_size, f0.voxel_size);
  uint2* cp = &m.color[(size_t)slot * 512 + tid];
  uint2 cur = make_uint2(0u, 0u);
  bool loaded = false, touched = false;
  // the cameras' blends are applied to the voxel in order, in registers: exactly what separate integrateColor calls would leave
#pragma unroll 1
  for (int c = 0; c < ncam; c++) {
    if (!((in_view >> c) & 1u)) continue;              // uniform
    const FrameCore& f = fs.f[c];
    const float* synth = synth_all + (size_t)c * srows * scols;
    float pc[3];
    apply_rt(f.R_CL, f.t_CL, lx, ly, lz, pc);
    float u, v;
    if (!cam_project(f, pc, &u, &v)) continue;
    const float vd = pc[2];
    if (f.max_dist > 0.0f && vd > f.max_dist) continue;
    // bilinear taps of the colour image (interpolate2DLinear<Color>) and of the synthetic depth: addresses first, then
    // all 4 + 12 loads in flight together
    const float uc = u - 0.5f, vc = v - 0.5f;
    const float fx = floorf(uc), fy = floorf(vc);
    const int x0 = (int)fx, y0 = (int)fy;
    const bool c_ok = !(x0 < 0 || y0 < 0 || x0 + 1 > f.cols - 1 || y0 + 1 > f.rows - 1);
    const float us = NVBX_DIV(u, (float)f.subsample), vs_ = NVBX_DIV(v, (float)f.subsample);
    const float usc = us - 0.5f, vsc = vs_ - 0.5f;
    const float sfx = floorf(usc), sfy = floorf(vsc);
    const int sx0 = (int)sfx, sy0 = (int)sfy;
    const bool s_ok = !(sx0 < 0 || sy0 < 0 || sx0 + 1 > scols - 1 || sy0 + 1 > srows - 1);
    if (!c_ok || !s_ok) continue;
    const float* sp = synth + pix(sy0, sx0, scols);
    const int32_t i00 = pix(y0, x0, f.cols);
    // (the colour voxel is only needed for the blend: it travels with the taps, not with the vote's inputs -- blocks
    // outside the truncation band or the frustum, most of the map, never fetch it)
    if (!loaded) { cur = *cp; loaded = true; }
    const float s00 = sp[0], s10 = sp[1], s01 = sp[scols], s11 = sp[scols + 1];
    float t00[3], t10[3], t01[3], t11[3];
    fs.img[c].tap(i00, t00); fs.img[c].tap(i00 + 1, t10); fs.img[c].tap(i00 + f.cols, t01); fs.img[c].tap(i00 + f.cols + 1, t11);
    // (the compiler sinks the colour taps below the occlusion test -- two round trips for a voxel that passes, none for the many that
    // fail; pinning them above it was measured: 9.5 -> 10.9 us)
    if (!(s00 > 0.0f) || !(s10 > 0.0f) || !(s01 > 0.0f) || !(s11 > 0.0f)) continue;
    const float sax = usc - sfx, say = vsc - sfy;
    const float stop = (1.0f - sax) * s00 + sax * s10;
    const float sbot = (1.0f - sax) * s01 + sax * s11;
    const float sd = (1.0f - say) * stop + say * sbot;
    if (fabsf(sd - vd) > f.occlusion_thresh) continue;      // [U] occlusion test (color_occlusion_threshold_vox)
    const float ax = uc - fx, ay = vc - fy;
    float cc[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float top = (1.0f - ax) * t00[ch] + ax * t10[ch];
      const float bot = (1.0f - ax) * t01[ch] + ax * t11[ch];
      cc[ch] = (1.0f - ay) * top + ay * bot;
    }
    const float w0 = __uint_as_float(cur.y);
    const uint32_t r8 = blend_u8((float)(cur.x & 0xFF), w0, cc[0], 1.0f);
    const uint32_t g8 = blend_u8((float)((cur.x >> 8) & 0xFF), w0, cc[1], 1.0f);
    const uint32_t b8 = blend_u8((float)((cur.x >> 16) & 0xFF), w0, cc[2], 1.0f);
    cur = make_uint2(r8 | (g8 << 8) | (b8 << 16), __float_as_uint(fminf(w0 + 1.0f, f.max_weight)));
    touched = true;
  }
  if (touched) *cp = cur;
}

// worker `wg` of `n_color_wg` 512-thread workgroups; every thread of the workgroup calls
template <typename Pix, int NB>
__device__ inline void color_integrate_worker(const DMap& m, const FrameSetC<Pix, NB>& fs, const float* synth_all, int32_t srows, int32_t scols, int32_t mesh_list,
                                              int32_t wg, int32_t n_color_wg) {
  __shared__ int s_out[NB][6];
  const int tid = threadIdx.x;
  const FrameCore& f0 = fs.f[0];
  const int ncam = NB > 1 ? fs.n : 1;
  // Candidate discovery.  Every allocated slot has to be looked at (O(map) flags, not O(view)): with one slot per workgroup iteration a
  // 10^5-block map costs ~150 dependent flag loads per workgroup.  So a workgroup takes `chunk` CONSECUTIVE slots per iteration (a
  // host hint, 1 .. 64, from the high-water mark the GPU last reported): lane l of every wavefront loads the flags and Index3D of slot
  // base + l, a ballot picks the blocks in the truncation band, and only those are visited.  chunk = 1 is the room-sized case: the
  // first slot's data is requested beside the high-water mark (indices clamped, so the addresses are valid).
  const int chunk = fs.chunk;
  const int lane_c = tid & 63;
  const int32_t cap = (int32_t)m.capacity;
  wg = xcd_chunked(wg, n_color_wg);       // runs of consecutive slots -- allocated together: neighbours, one patch of the colour image -- stay on one XCD's L2 (nvbx_internal.h)
  int32_t base = wg * chunk;
  int32_t ls = min(base + lane_c, cap - 1);
  uint32_t lflags = lane_c < chunk ? m.slot_flags[ls] : 0u;
  int32_t lbx = 0, lby = 0, lbz = 0;
  if (lane_c < chunk) { lbx = m.slot_index[3 * ls]; lby = m.slot_index[3 * ls + 1]; lbz = m.slot_index[3 * ls + 2]; }
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (; base < hw; base += n_color_wg * chunk) {
    if (base != wg * chunk) {
      ls = min(base + lane_c, cap - 1);
      lflags = lane_c < chunk ? m.slot_flags[ls] : 0u;
      if (lane_c < chunk) { lbx = m.slot_index[3 * ls]; lby = m.slot_index[3 * ls + 1]; lbz = m.slot_index[3 * ls + 2]; }
    }
    // the band vote ("any voxel with weight > 0 and |distance| < truncation") is the slot's F_BAND flag, kept exact by every
    // kernel that writes TSDF voxels (nvbx_internal.h): no TSDF read here -- unless a LiDAR scan left the block STALE, in which
    // case this workgroup votes from the TSDF once and repairs the bits
    // Every wavefront loads the chunk's flags on its own, and the STALE repair below REWRITES them (publish_band): all eight must have
    // their copy before the first of them publishes, or a lagging wavefront would see STALE already cleared and only part of the band bits,
    // pick a different candidate set and take different barriers (reachable only when LiDAR and colour share a mapper).
    __syncthreads();
    u64 cand = __ballot(lane_c < chunk && base + lane_c < hw && (lflags & F_TSDF) && (lflags & (F_BAND | F_BAND_STALE)));     // (the same in all eight wavefronts)
  while (cand) {
    const int cj = __ffsll((long long)cand) - 1;
    cand &= cand - 1ull;
    const int32_t slot = base + cj;
    const uint32_t flags = __shfl(lflags, cj);
    const int32_t bx = __shfl(lbx, cj), by = __shfl(lby, cj), bz = __shfl(lbz, cj);
    if (flags & F_BAND_STALE) {          // uniform
      const float2 tv = m.tsdf[(size_t)slot * 512 + tid];
      const bool pred = in_band(tv.x, tv.y, f0.trunc);
      publish_band(m.slot_flags, (uint32_t)slot, tid, pred);
      if (!__syncthreads_or(pred ? 1 : 0)) continue;
    }
    __syncthreads();
    if (tid < 6 * NB) (&s_out[0][0])[tid] = 0;
    __syncthreads();
    if (tid < 8 * ncam) {   // frustum: count corners outside each plane, 8 lanes per camera
      const int c = NB == 1 ? 0 : (tid >> 3), q = tid & 7;      // (one camera: a constant index -- a per-lane index into the argument block is a vector load from memory)
      const FrameCore& f = fs.f[c];
      float pc[3];
      apply_rt(f.R_CL, f.t_CL, (float)(bx + (q & 1)) * f.block_size, (float)(by + ((q >> 1) & 1)) * f.block_size,
               (float)(bz + ((q >> 2) & 1)) * f.block_size, pc);
      if (f.fu * pc[0] + f.cu * pc[2] < 0.0f) atomicAdd(&s_out[c][0], 1);
      if (f.fu * pc[0] + (f.cu - (float)f.w) * pc[2] > 0.0f) atomicAdd(&s_out[c][1], 1);
      if (f.fv * pc[1] + f.cv * pc[2] < 0.0f) atomicAdd(&s_out[c][2], 1);
      if (f.fv * pc[1] + (f.cv - (float)f.h) * pc[2] > 0.0f) atomicAdd(&s_out[c][3], 1);
      if (pc[2] < 0.0f) atomicAdd(&s_out[c][4], 1);
      if (f.max_dist > 0.0f && pc[2] > f.max_dist) atomicAdd(&s_out[c][5], 1);
    }
    __syncthreads();
    uint32_t in_view = 0u;               // cameras whose frustum the block touches (uniform)
    for (int c = 0; c < ncam; c++) {
      bool iv = true;
#pragma unroll
      for (int q = 0; q < 6; q++) if (s_out[c][q] == 8) iv = false;
      if (iv) in_view |= 1u << c;
    }
    if (!in_view) continue;                // uniform
    color_integrate_block<Pix, NB>(m, fs, synth_all, srows, scols, mesh_list, slot, bx, by, bz, in_view);
  }
  }
}

// ---- pipelined order with a fused colour + TSDF launch: the candidate discovery of colour frame i runs as a rider of the
// PREVIOUS launch (view marking of depth frame i + 1 -- nothing writes block flags or TSDF voxels there), so that the colour integration
// itself reads no flag the TSDF update of frame i + 1, running beside it, is changing (F_BAND).
// Candidate records {slot | camera mask << 24, block index}: exactly the blocks color_integrate_worker would visit -- TSDF layer, truncation-band
// flag, inside the frustum of at least one camera (the same expressions; the mask names which).  One wavefront per 64 consecutive slots
// (lane = slot), wave-aggregated append.  Slot ids stay below 2^24 (mapper.hip: max_capacity).
// Requires: no block flagged F_BAND_STALE (the host does not take this path once a LiDAR scan has been integrated into the mapper).
template <int NB>
__device__ inline void color_scan_worker(const DMap& m, const PoseSet<NB>& ps, int4* cand, int32_t cnt_idx, int32_t reset_idx, int w, int n_waves) {
  const int lane = threadIdx.x & 63;
  const int32_t cap = (int32_t)m.capacity;
  const int ncam = NB > 1 ? ps.n : 1;
  NVBX_INV_TSDF_READER(m);
  if (w == 0 && lane == 0) m.counters[reset_idx] = 0;       // the other parity's count: consumed one launch ago, appended to by the next scan
  int32_t base = w * 64;
  int32_t s = min(base + lane, cap - 1);
  uint32_t flags = m.slot_flags[s];                         // (speculative, beside the high-water mark)
  int32_t bx = m.slot_index[3 * s], by = m.slot_index[3 * s + 1], bz = m.slot_index[3 * s + 2];
  const int32_t hw = m.counters[C_HIGH_WATER];
  for (; base < hw; base += n_waves * 64) {
    if (base != w * 64) {
      s = min(base + lane, cap - 1);
      flags = m.slot_flags[s]; bx = m.slot_index[3 * s]; by = m.slot_index[3 * s + 1]; bz = m.slot_index[3 * s + 2];
    }
    uint32_t in_view = 0u;
    if (base + lane < hw && (flags & F_TSDF) && (flags & F_BAND)) {
#pragma unroll 1
      for (int c = 0; c < ncam; c++) {          // frustum: corners outside each plane (color_integrate_worker's test, one lane per block)
        const FrameCore& f = ps.f[NB == 1 ? 0 : c];
        int out[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 8; q++) {
          float pc[3];
          apply_rt(f.R_CL, f.t_CL, (float)(bx + (q & 1)) * f.block_size, (float)(by + ((q >> 1) & 1)) * f.block_size, (float)(bz + ((q >> 2) & 1)) * f.block_size, pc);
          if (f.fu * pc[0] + f.cu * pc[2] < 0.0f) out[0]++;
          if (f.fu * pc[0] + (f.cu - (float)f.w) * pc[2] > 0.0f) out[1]++;
          if (f.fv * pc[1] + f.cv * pc[2] < 0.0f) out[2]++;
          if (f.fv * pc[1] + (f.cv - (float)f.h) * pc[2] > 0.0f) out[3]++;
          if (pc[2] < 0.0f) out[4]++;
          if (f.max_dist > 0.0f && pc[2] > f.max_dist) out[5]++;
        }
        bool iv = true;
#pragma unroll
        for (int q = 0; q < 6; q++) if (out[q] == 8) iv = false;
        if (iv) in_view |= 1u << c;
      }
    }
    const bool keep = in_view != 0u;
    const u64 km = __ballot(keep);
    if (km) {
      int32_t pos0 = 0;
      if (lane == 0) pos0 = atomicAdd(&m.counters[cnt_idx], (int32_t)__popcll(km));
      pos0 = __shfl(pos0, 0);
      const int32_t pos = pos0 + (int32_t)__popcll(km & ((1ull << lane) - 1ull));
      if (keep && pos < cap) cand[pos] = make_int4((int32_t)((uint32_t)s | (in_view << 24)), bx, by, bz);
    }
  }
}

// worker `wg` of `n_wg` 512-thread workgroups over the candidate records
template <typename Pix, int NB>
__device__ inline void color_integrate_list_worker(const DMap& m, const FrameSetC<Pix, NB>& fs, const float* synth, int32_t srows, int32_t scols, int32_t mesh_list,
                                                   const int4* cand, int32_t cnt_idx, int32_t wg, int32_t n_wg) {
  const int32_t cap = (int32_t)m.capacity;
  const int32_t wg0 = wg;
  wg = xcd_chunked(wg, n_wg);                               // runs of consecutive candidates (neighbouring blocks: one patch of the colour image) stay on one XCD's L2
  int4 rec = cand[min(wg, cap - 1)];                        // (speculative, beside the count)
  int32_t n = m.counters[cnt_idx];
  if (n > cap) n = cap;
  if (wg0 == 0 && threadIdx.x == 0) __hip_atomic_store(&m.host_mirror[3], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // the next fused launch's grid hint
  for (int32_t i = wg; i < n; i += n_wg) {
    if (i != wg) rec = cand[i];
    NVBX_INV_COUNT(m, C_INV_I3, threadIdx.x == 0 && (m.slot_index[3 * ((uint32_t)rec.x & 0xFFFFFFu)] != rec.y || m.slot_index[3 * ((uint32_t)rec.x & 0xFFFFFFu) + 1] != rec.z || m.slot_index[3 * ((uint32_t)rec.x & 0xFFFFFFu) + 2] != rec.w));
    color_integrate_block<Pix, NB>(m, fs, synth, srows, scols, mesh_list, (int32_t)((uint32_t)rec.x & 0xFFFFFFu), rec.y, rec.z, rec.w, (uint32_t)rec.x >> 24);
  }
}

}  // namespace nvbx
