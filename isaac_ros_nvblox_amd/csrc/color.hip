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
#include <cstdlib>
#include <cstring>
#include "nvbx_mapper.h"
#include "nvbx_esdf_mark.h"
#include "nvbx_sphere_trace.h"

using namespace nvbx;

template <typename Pix, int NB> struct FrameSetC { Frame f[NB]; Pix img[NB]; int32_t n; int32_t chunk; };

template <int NB, int RAY_LANES>
__global__ __launch_bounds__(256) void k_sphere_trace(DMap m, PoseSet<NB> poses, float* synth_all, int32_t srows, int32_t scols, int32_t max_steps,
                                                      float max_len, float eps_m) {
  sphere_trace_worker<NB, RAY_LANES>(m, poses, synth_all, srows, scols, max_steps, max_len, eps_m, (int)blockIdx.x);
}

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

// Dependent-access chain: {slot flags (incl. the exact band flag), Index3D} (addressed by the slot id alone, fetched together)
// -> frustum vote -> {synthetic depth gather, colour gather, colour voxel} (fetched together) -> store.
// Workgroups [0, n_mark_wg) are ESDF marking workers (first wavefront only; dispatched first so that they start at once and
// do not queue for a CU slot behind the resident batch of colour workgroups); the n_color_wg after them integrate colour.
// (7 waves/SIMD at 67 VGPRs; forcing the eighth -- amdgpu_waves_per_eu(8, 8), 63 VGPRs -- was measured: 9.5 -> 10.3 us, rejected)
template <typename Pix, int NB>
__global__ __launch_bounds__(512) void k_integrate_color(DMap m, FrameSetC<Pix, NB> fs, const float* synth_all, int32_t srows, int32_t scols,
                                                         int32_t mesh_list, int32_t n_mark_wg, EsdfArgs ea, ImportArgs imp) {
  if ((int32_t)blockIdx.x < n_mark_wg) {
    if (threadIdx.x < 64) {
      // workers [0, n_own) re-mark the mapper's own dirty blocks, workers [n_own, n_mark_wg) the peers' gathered lists
      // (multi-GPU union step held back by nvbx_mark_esdf_dirty_gathered_deferred); one marking pass, one column stamp
      const int32_t n_own = n_mark_wg - imp.n_wg;
      if ((int32_t)blockIdx.x < n_own) { esdf_mark_worker(m, ea, (int)blockIdx.x, n_own); esdf_mark_pass_done(m, ea, n_own); }
      else esdf_import_mark_worker(m, ea, imp, (int)blockIdx.x - n_own);
    }
    return;
  }
  const int32_t wg = (int32_t)blockIdx.x - n_mark_wg, n_color_wg = (int32_t)gridDim.x - n_mark_wg;
  __shared__ int s_out[NB][6];
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  const Frame& f0 = fs.f[0];
  const int ncam = NB > 1 ? fs.n : 1;
  // Candidate discovery.  Every allocated slot has to be looked at (O(map) flags, not O(view)): with one slot per workgroup iteration a
  // 10^5-block map costs ~150 dependent flag loads per workgroup.  So a workgroup takes `chunk` CONSECUTIVE slots per iteration (a
  // host hint, 1 .. 64, from the high-water mark the GPU last reported): lane l of every wavefront loads the flags and Index3D of slot
  // base + l, a ballot picks the blocks in the truncation band, and only those are visited.  chunk = 1 is the room-sized case: the
  // first slot's data is requested beside the high-water mark (indices clamped, so the addresses are valid).
  const int chunk = fs.chunk;
  const int lane_c = tid & 63;
  const int32_t cap = (int32_t)m.capacity;
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
      const Frame& f = fs.f[c];
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
      const Frame& f = fs.f[c];
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
  }
}

// lanes per ray of the sphere-tracing launch for a batch of n cameras (1, 2, 4 or 8)
static int sphere_trace_lanes(int n) {
  static const int forced = getenv("NVBX_ST_LANES") ? atoi(getenv("NVBX_ST_LANES")) : 0;       // (sweeps)
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
  // measured (tools/st_lanes_sweep.sh, profiles/r03d_st_lanes.txt; us per launch at 8 / 4 / 2 / 1 lanes): 8 cameras 30.6 / 23.6 / 21.8 / 26.3,
  // 4 cameras 20.0 / 16.7 / 18.2 / 24.4, 2 cameras 13.1 / 13.5 / 16.1 / 21.4
  return n >= 6 ? 2 : (n >= 3 ? 4 : 8);
}

// n colour frames (n = 1: MultiMapper::integrateColor; n > 1: nvbx_integrate_color_batch) of one image size -> one launch set, in three
// steps so that a held-back frame (colour deferral, nvbx_mapper.h) can have its sphere tracing launched by the next depth frame:
// color_setup (checks, frames, scratch), color_launch_trace, color_launch_integrate.
static int color_precheck(nvbx_mapper* m, int32_t n, int32_t rows, int32_t cols, const float* T_L_C) {
  for (int c = 0; c < n; c++)
    if (!nvbx_pose_in_range(T_L_C + 16 * c, m->p.voxel_size * 8.0f, m->p.sphere_tracing_max_ray_length_m + m->p.max_integration_distance_m)) {
      set_error("integrate color: T_L_C is not finite or lies outside the addressable block range (+-2^20 blocks)"); return NVBX_E_INVALID; }
  const int sub = std::max(1, m->p.sphere_tracing_subsampling);
  if (rows / sub < 2 || cols / sub < 2) { set_error("colour image too small for the sphere-tracing subsampling"); return NVBX_E_INVALID; }
  return NVBX_OK;
}
template <typename Pix, int NB>
static int color_setup(nvbx_mapper* m, int32_t n, const Pix* imgs, int32_t rows, int32_t cols, const float* T_L_C, const nvbx_camera* cameras,
                       FrameSetC<Pix, NB>* fs, PoseSet<NB>* ps, int32_t* srows_out, int32_t* scols_out) {
  *fs = FrameSetC<Pix, NB>{}; fs->n = n;
  *ps = PoseSet<NB>{}; ps->n = n;
  for (int c = 0; c < n; c++) { fs->f[c] = m->make_frame(T_L_C + 16 * c, cameras + c, rows, cols, m->p.sphere_tracing_subsampling); fs->img[c] = imgs[c]; ps->f[c] = fs->f[c]; }
  const Frame& f = fs->f[0];
  { // slots per workgroup iteration of k_integrate_color's candidate scan: from the high-water mark the GPU last reported (a hint only)
    const int64_t hw_seen = std::max<int64_t>(1, __atomic_load_n(&m->h_mirror[1], __ATOMIC_RELAXED));
    int ch = 1; while (ch < 64 && (int64_t)ch * std::min<int64_t>(m->capacity, 1024) * 4 < hw_seen) ch *= 2;     // (up to 4 iterations of single slots: a room-sized map keeps
                                                                                                        //  the stride-grid pairing, which balances the in-band blocks better than runs of neighbours)
    fs->chunk = ch; }
  const int32_t srows = rows / f.subsample, scols = cols / f.subsample;
  if (srows < 2 || scols < 2) { set_error("colour image too small for the sphere-tracing subsampling"); return NVBX_E_INVALID; }
  if ((int64_t)srows * scols * n > m->synth_cap) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->synth) NVBX_HIP(hipFree(m->synth));
    m->synth = nullptr; m->synth_cap = 0;
    NVBX_HIP(hipMalloc(&m->synth, (size_t)srows * scols * n * 4));
    m->synth_cap = (int64_t)srows * scols * n;
  }
  m->synth_rows = srows; m->synth_cols = scols; m->synth_last = n - 1;
  *srows_out = srows; *scols_out = scols;
  return NVBX_OK;
}
// workgroups of the sphere-tracing launch / rider: one 256-thread workgroup per 8 x (32 / lanes) patch of rays, NSH x ceil(patches / NSH)
// per camera (XCD-banded numbering, see the worker)
static int sphere_trace_workgroups(int rl, int32_t srows, int32_t scols, int n) {
  const int ph = 256 / rl / 8;
  const int st_patches = ((scols + 7) / 8) * ((srows + ph - 1) / ph);
  return NSH * ((st_patches + NSH - 1) / NSH) * n;
}
template <int NB>
static int color_launch_trace(nvbx_mapper* m, const PoseSet<NB>& ps, int32_t n, int32_t srows, int32_t scols) {
  const int rl = sphere_trace_lanes(n);
  const dim3 st_grid((unsigned)sphere_trace_workgroups(rl, srows, scols, n));
#define NVBX_ST_LAUNCH(RL) NVBX_LAUNCH(m, (k_sphere_trace<NB, RL>), st_grid, dim3(256), m->d, ps, m->synth, srows, scols, m->p.sphere_tracing_max_steps, \
                                       m->p.sphere_tracing_max_ray_length_m, m->p.sphere_tracing_surface_eps_vox * m->p.voxel_size)
  if (rl == 8) NVBX_ST_LAUNCH(8); else if (rl == 4) NVBX_ST_LAUNCH(4); else if (rl == 2) NVBX_ST_LAUNCH(2); else NVBX_ST_LAUNCH(1);
#undef NVBX_ST_LAUNCH
  return NVBX_OK;
}
template <typename Pix, int NB>
static int color_launch_integrate(nvbx_mapper* m, const FrameSetC<Pix, NB>& fs, int32_t srows, int32_t scols) {
  const int grid = (int)std::min<int64_t>(m->capacity, 1024);     // one resident batch of 512-thread workgroups
  // ESDF site marking of the blocks dirtied since the last marking pass rides in this launch (256 extra single-wavefront
  // workers): it reads only the TSDF, like the colour pass, and a following updateEsdf then needs the EDT kernel only
  int mark_wg = 0;
  EsdfArgs ea = m->make_esdf_args();
  ImportArgs imp{};
  const bool own = m->p.esdf_mode == 0 && m->p.esdf_propagation == 0 && m->dirty_since_mark && !m->premark_consumed && ea.bz_hi >= ea.bz_lo && ea.bz_hi - ea.bz_lo + 1 <= 63;
  if (own || m->import_pending) {
    m->mark_pass++; ea.mark_pass = m->mark_pass; m->unresolved_marks = true;
    if (own) { mark_wg = 256; m->dirty_since_mark = false; m->premark_consumed = !ea.self_reset; }     // (pipelined order: the pass empties the list itself)
    if (m->import_pending) {             // (only ever set in 2-D mode with a valid band: nvbx_mark_esdf_dirty_gathered_deferred)
      imp.g = m->import_ptr; imp.world = m->import_world; imp.self_rank = m->import_rank; imp.max_count = m->import_max;
      { const int n_peers = std::max(1, (m->import_rank >= 0 && m->import_rank < m->import_world) ? m->import_world - 1 : m->import_world);
        imp.n_wg = n_peers * std::max(32, std::min(256, 1024 / n_peers)); }       // ~one list entry per worker at a few hundred blocks per peer
      mark_wg += imp.n_wg; m->import_pending = false;
    }
  }
  NVBX_LAUNCH(m, (k_integrate_color<Pix, NB>), dim3(grid + mark_wg), dim3(512), m->d, fs, m->synth, srows, scols, m->mesh_list_live(), (int32_t)mark_wg, ea, imp);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
template <typename Pix, int NB>
static int integrate_colors(nvbx_mapper* m, int32_t n, const Pix* imgs, int32_t rows, int32_t cols, const float* T_L_C /* n x 16 */, const nvbx_camera* cameras) {
  { const int rc = color_precheck(m, n, rows, cols, T_L_C); if (rc) return rc; }
  NVBX_HIP(hipSetDevice(m->device));
  if (!m->replaying && m->replay_deferred()) return NVBX_E_DEVICE;      // an older held-back colour frame / ESDF update goes first
  if (m->p.projective_layer_type == 1) return NVBX_OK;      // occupancy mappers carry no colour (the occlusion test sphere-traces a TSDF)
  if (m->flush_edt()) return NVBX_E_DEVICE;      // a held-back EDT must precede this launch's marking pass (it reads the site masks)
  FrameSetC<Pix, NB> fs; PoseSet<NB> ps; int32_t srows = 0, scols = 0;
  { const int rc = color_setup<Pix, NB>(m, n, imgs, rows, cols, T_L_C, cameras, &fs, &ps, &srows, &scols); if (rc) return rc; }
  { const int rc = color_launch_trace<NB>(m, ps, n, srows, scols); if (rc) return rc; }
  return color_launch_integrate<Pix, NB>(m, fs, srows, scols);
}

// ---- colour deferral (nvbx_mapper.h): hold a single frame back / carry it out in pipelined order
static bool defer_color(nvbx_mapper* m, int kind, const void* img, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera, int* rc_out) {
  if (!m->color_deferral || m->replaying || m->p.projective_layer_type == 1) return false;
  *rc_out = color_precheck(m, 1, rows, cols, T_L_C);            // argument errors are reported by the call that made them
  if (*rc_out) return true;
  if (hipSetDevice(m->device) != hipSuccess || m->replay_deferred()) { *rc_out = NVBX_E_DEVICE; return true; }     // an older held-back frame goes first
  nvbx_mapper::ColorPending& c = m->color_pending;
  c.on = true; c.kind = kind; c.img = img; c.rows = rows; c.cols = cols; memcpy(c.T, T_L_C, sizeof(c.T)); c.cam = *camera;
  *rc_out = NVBX_OK;
  return true;
}
// the held-back frame's set-up; its sphere tracing as a rider of the caller's launch
static_assert(sizeof(FrameSetC<PixRgb8, 1>) == sizeof(FrameSetC<PixBgra8, 1>), "colour frame sets share one layout");
int nvbx_mapper::pending_color_trace_rider(void* out) {
  TraceRider* tr = static_cast<TraceRider*>(out);
  const ColorPending& c = color_pending;
  int32_t srows = 0, scols = 0;
  int rc;
  if (c.kind == 0) { const PixRgb8 img{(const uint8_t*)c.img}; FrameSetC<PixRgb8, 1> fs; rc = color_setup<PixRgb8, 1>(this, 1, &img, c.rows, c.cols, c.T, &c.cam, &fs, &tr->ps, &srows, &scols); }
  else { const PixBgra8 img{(const uint32_t*)c.img}; FrameSetC<PixBgra8, 1> fs; rc = color_setup<PixBgra8, 1>(this, 1, &img, c.rows, c.cols, c.T, &c.cam, &fs, &tr->ps, &srows, &scols); }
  if (rc) return rc;
  tr->synth = synth; tr->srows = srows; tr->scols = scols; tr->max_steps = p.sphere_tracing_max_steps;
  tr->max_len = p.sphere_tracing_max_ray_length_m; tr->eps_m = p.sphere_tracing_surface_eps_vox * p.voxel_size;
  static const int fused_lanes = getenv("NVBX_FUSED_TRACE_LANES") ? atoi(getenv("NVBX_FUSED_TRACE_LANES")) : 8;       // (A/B: 4 or 8 lanes per ray)
  tr->lanes = fused_lanes == 4 ? 4 : 8;
  tr->n_wg = sphere_trace_workgroups(tr->lanes, srows, scols, 1);
  return NVBX_OK;
}
int nvbx_mapper::launch_pending_color_after_trace() {
  const ColorPending c = color_pending; color_pending.on = false;
  int32_t srows = 0, scols = 0;
  if (c.kind == 0) {
    const PixRgb8 img{(const uint8_t*)c.img}; FrameSetC<PixRgb8, 1> fs; PoseSet<1> ps;
    const int rc = color_setup<PixRgb8, 1>(this, 1, &img, c.rows, c.cols, c.T, &c.cam, &fs, &ps, &srows, &scols); if (rc) return rc;
    return color_launch_integrate<PixRgb8, 1>(this, fs, srows, scols);
  }
  const PixBgra8 img{(const uint32_t*)c.img}; FrameSetC<PixBgra8, 1> fs; PoseSet<1> ps;
  const int rc = color_setup<PixBgra8, 1>(this, 1, &img, c.rows, c.cols, c.T, &c.cam, &fs, &ps, &srows, &scols); if (rc) return rc;
  return color_launch_integrate<PixBgra8, 1>(this, fs, srows, scols);
}

extern "C" int nvbx_integrate_color(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                    const nvbx_camera* camera) {
  if (!m || !rgb_dev || !T_L_C || !camera || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_color: invalid argument (image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_color: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  { int rc = NVBX_OK; if (defer_color(m, 0, rgb_dev, rows, cols, T_L_C, camera, &rc)) return rc; }
  const PixRgb8 img{rgb_dev};
  return integrate_colors<PixRgb8, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
extern "C" int nvbx_integrate_color_bgra8(nvbx_mapper* m, const uint8_t* bgra_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                          const nvbx_camera* camera) {
  if (!m || !bgra_dev || !T_L_C || !camera || !image_dims_ok(rows, cols) || ((uintptr_t)bgra_dev & 3)) { set_error("nvbx_integrate_color_bgra8: invalid argument"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_color_bgra8: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  { int rc = NVBX_OK; if (defer_color(m, 1, bgra_dev, rows, cols, T_L_C, camera, &rc)) return rc; }
  const PixBgra8 img{reinterpret_cast<const uint32_t*>(bgra_dev)};
  return integrate_colors<PixBgra8, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
// Up to NVBX_MAX_BATCH colour frames (rgb8, same image size) in ONE launch set: see include/nvblox_hip.h
extern "C" int nvbx_integrate_color_batch(nvbx_mapper* m, int32_t n, const uint8_t* const* rgb_dev, int32_t rows, int32_t cols, const float* T_L_C,
                                          const nvbx_camera* cameras) {
  if (!m || n < 1 || n > MAX_BATCH || !rgb_dev || !T_L_C || !cameras || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_color_batch: invalid argument (1 <= n <= 8, image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  for (int c = 0; c < n; c++)
    if (!rgb_dev[c] || !nvbx_camera_matches(cameras + c, rows, cols)) { set_error("nvbx_integrate_color_batch: every camera's width/height must equal the images' cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  if (n == 1) return nvbx_integrate_color(m, rgb_dev[0], rows, cols, T_L_C, cameras);
  PixRgb8 imgs[MAX_BATCH];
  for (int c = 0; c < n; c++) imgs[c] = PixRgb8{rgb_dev[c]};
  return integrate_colors<PixRgb8, MAX_BATCH>(m, n, imgs, rows, cols, T_L_C, cameras);
}

extern "C" int nvbx_get_synthetic_depth(nvbx_mapper* m, float* out_host, int64_t capacity, int32_t* rows, int32_t* cols) {
  if (!m || !rows || !cols) return NVBX_E_INVALID;
  if (m->join_side()) return NVBX_E_DEVICE;          // (a held-back colour frame is carried out first)
  *rows = m->synth_rows; *cols = m->synth_cols;
  const int64_t n = (int64_t)m->synth_rows * m->synth_cols;
  if (!out_host || n == 0) return NVBX_OK;
  if (n > capacity) return NVBX_E_CAPACITY;
  NVBX_HIP(hipMemcpyAsync(out_host, m->synth + (size_t)m->synth_last * n, n * 4, hipMemcpyDeviceToHost, m->stream));    // (a batch: its last camera's image)
  NVBX_HIP(hipStreamSynchronize(m->stream));
  return NVBX_OK;
}
