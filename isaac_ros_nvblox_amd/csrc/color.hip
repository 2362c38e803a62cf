// color.hip -- MultiMapper::integrateColor on MI355X.
//
// Classic order, two launches per colour frame (the workers live in headers so that other launches can carry them):
//   k_sphere_trace    sample-parallel sphere tracing of the 1/f-resolution synthetic depth image, 8 lanes per ray
//                     (nvbx_sphere_trace.h; [U] SphereTracer::cast restated)
//   k_integrate_color one 512-thread workgroup per allocated block slot (grid-stride over the slot range): truncation-band flag,
//                     frustum test (8 lanes = 8 corners), then the per-voxel projective colour blend -- selection and integration
//                     fused, no block list round trip -- with the ESDF site marking riding in extra workgroups
//                     (nvbx_color_worker.h; [U] ProjectiveColorIntegrator::integrateFrame restated).
// Pipelined order (colour deferral): the frame is held back; its sphere tracing, candidate discovery and the marking
// pass ride in the next depth frame's view-marking launch and its colour integration in that frame's TSDF-update launch (tsdf.hip) --
// this file keeps the held-back state's set-up (pending_*) and the replay in classic order.
// Call site served: nvblox_ros/src/lib/nvblox_node.cpp:1264.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "nvbx_mapper.h"
#include "nvbx_esdf_mark.h"
#include "nvbx_esdf_edt.h"
#include "nvbx_sphere_trace.h"
#include "nvbx_color_worker.h"

using namespace nvbx;


// (n_mark_wg > 0: behind the n_trace_wg sphere-tracing workgroups, the ESDF site marking of a held-back updateEsdf -- first wavefront only; both
//  read the TSDF only.  Used when a held-back colour frame AND its updateEsdf are replayed together: nvbx_mapper::replay_pair)
template <int NB, int RAY_LANES>
__global__ __launch_bounds__(256) void k_sphere_trace(DMap m, PoseSet<NB> poses, float* synth_all, int32_t srows, int32_t scols, int32_t max_steps,
                                                      float max_len, float eps_m, int32_t n_trace_wg, int32_t n_mark_wg, EsdfArgs ea) {
  if ((int32_t)blockIdx.x >= n_trace_wg) {
    if (threadIdx.x < 64) { const int w = (int)blockIdx.x - n_trace_wg; esdf_mark_worker(m, ea, w, n_mark_wg); esdf_mark_pass_done(m, ea, n_mark_wg, w); }
    return;
  }
  sphere_trace_worker<NB, RAY_LANES>(m, poses, synth_all, srows, scols, max_steps, max_len, eps_m, (int)blockIdx.x);
}

// Dependent-access chain: {slot flags (incl. the exact band flag), Index3D} (addressed by the slot id alone, fetched together)
// -> frustum vote -> {synthetic depth gather, colour gather, colour voxel} (fetched together) -> store.
// Workgroups [0, n_mark_wg) are ESDF marking workers (first wavefront only; dispatched first so that they start at once and
// do not queue for a CU slot behind the resident batch of colour workgroups); the n_color_wg after them integrate colour.
// (7 waves/SIMD at 67 VGPRs; forcing the eighth -- amdgpu_waves_per_eu(8, 8), 63 VGPRs -- was measured: 9.5 -> 10.3 us, rejected)
template <typename Pix, int NB>
__global__ __launch_bounds__(512) void k_integrate_color(DMap m, FrameSetC<Pix, NB> fs, const float* synth_all, int32_t srows, int32_t scols,
                                                         int32_t mesh_list, int32_t n_mark_wg, EsdfArgs ea, ImportArgs imp, int32_t n_edt_wg) {
  // (n_edt_wg > 0, then n_mark_wg = 0: the distance transform of the held-back updateEsdf whose marking pass rode in the sphere-tracing launch
  //  -- nvbx_mapper::replay_pair; it touches the ESDF layer only; `ea` is its argument)
  __shared__ __align__(16) unsigned char edt_smem[sizeof(EdtShared)];
  if ((int32_t)blockIdx.x < n_edt_wg) { esdf_edt_worker<512>(m, ea, (int)blockIdx.x, n_edt_wg, reinterpret_cast<EdtShared*>(edt_smem)); return; }
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
  color_integrate_worker<Pix, NB>(m, fs, synth_all, srows, scols, mesh_list, (int32_t)blockIdx.x - n_mark_wg - n_edt_wg, (int32_t)gridDim.x - n_mark_wg - n_edt_wg);
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
  const FrameCore& f = fs->f[0];
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
static int color_launch_trace(nvbx_mapper* m, const PoseSet<NB>& ps, int32_t n, int32_t srows, int32_t scols, int32_t mark_wg = 0, const EsdfArgs& ea = EsdfArgs{}) {
  const int rl = sphere_trace_lanes(n);
  const int32_t n_trace = sphere_trace_workgroups(rl, srows, scols, n);
  const dim3 st_grid((unsigned)(n_trace + mark_wg));
#define NVBX_ST_LAUNCH(RL) NVBX_LAUNCH(m, (k_sphere_trace<NB, RL>), st_grid, dim3(256), m->d, ps, m->synth, srows, scols, m->p.sphere_tracing_max_steps, \
                                       m->p.sphere_tracing_max_ray_length_m, m->p.sphere_tracing_surface_eps_vox * m->p.voxel_size, n_trace, mark_wg, ea)
  if (rl == 8) NVBX_ST_LAUNCH(8); else if (rl == 4) NVBX_ST_LAUNCH(4); else if (rl == 2) NVBX_ST_LAUNCH(2); else NVBX_ST_LAUNCH(1);
#undef NVBX_ST_LAUNCH
  return NVBX_OK;
}
template <typename Pix, int NB>
static int color_launch_integrate(nvbx_mapper* m, const FrameSetC<Pix, NB>& fs, int32_t srows, int32_t scols, bool take_edt = false) {
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
  // (replay_pair: the distance transform the held-back updateEsdf has just armed rides here -- its marking pass rode in the sphere-tracing launch)
  int32_t n_edt = 0;
  if (take_edt && mark_wg == 0 && m->edt_pending) { n_edt = 256; ea = m->edt_args; m->edt_pending = false; }
  NVBX_LAUNCH(m, (k_integrate_color<Pix, NB>), dim3(grid + mark_wg + n_edt), dim3(512), m->d, fs, m->synth, srows, scols, m->mesh_list_live(), (int32_t)mark_wg, ea, imp, n_edt);
  NVBX_HIP(hipGetLastError());
  // this launch looks at every allocated slot and votes + repairs the band bits of each block a LiDAR scan left F_BAND_STALE: from here on (stream
  // order) no block is stale, and the next camera frames may take the fused launches again (the switch was sticky until clear())
  m->lidar_integrated = false;
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

// ---- colour deferral (nvbx_mapper.h): hold a frame (or a batch) back / carry it out in pipelined order
// the staged form's copy of the held-back images (one launch for a whole batch; the runtime's own device-to-device copy is a launch per image, ~4 us each)
struct StagePtrs { const unsigned char* src[MAX_BATCH]; unsigned char* dst[MAX_BATCH]; };
__global__ __launch_bounds__(256) void k_stage_color(StagePtrs p, int64_t bytes, int32_t wg_per_image) {
  const int img = (int)blockIdx.x / wg_per_image, wg = (int)blockIdx.x - img * wg_per_image;
  const unsigned char* s = p.src[img]; unsigned char* d = p.dst[img];
  if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0) {
    const int64_t n16 = bytes >> 4;
    for (int64_t i = (int64_t)wg * 256 + threadIdx.x; i < n16; i += (int64_t)wg_per_image * 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    if (wg == 0 && (int64_t)threadIdx.x < (bytes & 15)) d[(n16 << 4) + threadIdx.x] = s[(n16 << 4) + threadIdx.x];
  } else {
    for (int64_t i = (int64_t)wg * 256 + threadIdx.x; i < bytes; i += (int64_t)wg_per_image * 256) d[i] = s[i];
  }
}
static bool defer_color(nvbx_mapper* m, int kind, int32_t n, const void* const* imgs, int32_t rows, int32_t cols, const float* T_L_C, const nvbx_camera* cameras, int* rc_out) {
  if (!m->color_deferral || m->replaying || m->p.projective_layer_type == 1) return false;
  *rc_out = color_precheck(m, n, rows, cols, T_L_C);            // argument errors are reported by the call that made them
  if (*rc_out) return true;
  if (hipSetDevice(m->device) != hipSuccess || m->replay_deferred()) { *rc_out = NVBX_E_DEVICE; return true; }     // an older held-back frame goes first
  m->release_consumed_frames();
  nvbx_mapper::ColorPending& c = m->color_pending;
  const size_t bytes = (size_t)rows * (size_t)cols * (kind == 0 ? 3u : 4u);
  // Where each held-back image lives until it is carried out:
  //   a frame of nvbx_frame_acquire (nvblox::Image<T> device memory, nvbx_color_image_acquire): RETAINED -- no copy, in either mode (frames.hip);
  //   a raw device pointer, staged form (the default): copied into a frame of the mapper's own, one launch for the whole call (the caller may
  //     recycle or overwrite its image as soon as integrateColor has returned -- ADVICE r03);
  //   a raw device pointer, zero-copy form (opt-in): used where it lies, under the caller's contract (nvblox_hip.h).
  void* frames[MAX_BATCH] = {}; const void* use[MAX_BATCH] = {};
  StagePtrs sp{}; int n_copy = 0;
  auto undo = [&]() { for (int i = 0; i < n; i++) if (frames[i]) (void)nvbx_frame_release(frames[i]); };
  for (int i = 0; i < n; i++) {
    bool too_small = false;
    if (frame_retain_if_frame(imgs[i], bytes, &too_small)) { frames[i] = const_cast<void*>(imgs[i]); use[i] = imgs[i]; continue; }
    if (too_small) { undo(); set_error("integrate color: the frame of nvbx_frame_acquire is smaller than rows x cols pixels"); *rc_out = NVBX_E_INVALID; return true; }
    if (!m->color_staging) { use[i] = imgs[i]; continue; }
    void* f = nullptr;
    if (nvbx_frame_acquire(m->device, bytes, m->stream, &f) != NVBX_OK) { undo(); *rc_out = NVBX_E_DEVICE; return true; }
    frames[i] = f; use[i] = f;
    sp.src[n_copy] = (const unsigned char*)imgs[i]; sp.dst[n_copy] = (unsigned char*)f; n_copy++;
  }
  if (n_copy) {
    const int32_t wgi = (int32_t)std::min<int64_t>(256, std::max<int64_t>(1, (int64_t)(bytes >> 4) / 256 + 1));      // 640x480 rgb8: 226 workgroups, one 16-B access per thread
    NVBX_LAUNCH(m, k_stage_color, dim3((unsigned)(wgi * n_copy)), dim3(256), sp, (int64_t)bytes, wgi);
    if (hipGetLastError() != hipSuccess) { undo(); set_error("colour staging copy"); *rc_out = NVBX_E_DEVICE; return true; }
  }
  c.on = true; c.kind = kind; c.n = n; c.rows = rows; c.cols = cols;
  for (int i = 0; i < MAX_BATCH; i++) { c.imgs[i] = i < n ? use[i] : nullptr; c.frames[i] = i < n ? frames[i] : nullptr; if (i < n) c.cams[i] = cameras[i]; }
  memcpy(c.T, T_L_C, sizeof(float) * 16 * (size_t)n);
  *rc_out = NVBX_OK;
  return true;
}
// the held-back frames' set-up (NB = 1: one frame, NB = MAX_BATCH: a batch)
static_assert(sizeof(FrameSetC<PixRgb8, 1>) == sizeof(FrameSetC<PixBgra8, 1>), "colour frame sets share one layout");
template <typename Pix, int NB>
static int pending_setup(nvbx_mapper* m, const nvbx_mapper::ColorPending& c, FrameSetC<Pix, NB>* fs, PoseSet<NB>* ps, int32_t* srows, int32_t* scols) {
  Pix imgs[NB];
  for (int i = 0; i < c.n && i < NB; i++) imgs[i] = Pix{reinterpret_cast<decltype(Pix::p)>(c.imgs[i])};
#ifdef NVBX_CHECK_INVARIANTS
  for (int i = 0; i < c.n && i < NB; i++) if (c.frames[i] && nvbx_frame_refcount(c.frames[i]) < 1) m->inv_i8++;      // I8: a held-back image's frame is held until its readers are enqueued
#endif
  return color_setup<Pix, NB>(m, c.n, imgs, c.rows, c.cols, c.T, c.cams, fs, ps, srows, scols);
}
template <int NB>
static int trace_rider_of(nvbx_mapper* m, TraceRiderT<NB>* tr) {
  const nvbx_mapper::ColorPending& c = m->color_pending;
  int32_t srows = 0, scols = 0;
  int rc;
  if (c.kind == 0) { FrameSetC<PixRgb8, NB> fs; rc = pending_setup<PixRgb8, NB>(m, c, &fs, &tr->ps, &srows, &scols); }
  else { FrameSetC<PixBgra8, NB> fs; rc = pending_setup<PixBgra8, NB>(m, c, &fs, &tr->ps, &srows, &scols); }
  if (rc) return rc;
  tr->synth = m->synth; tr->srows = srows; tr->scols = scols; tr->max_steps = m->p.sphere_tracing_max_steps;
  tr->max_len = m->p.sphere_tracing_max_ray_length_m; tr->eps_m = m->p.sphere_tracing_surface_eps_vox * m->p.voxel_size;
  static const int fused_lanes = getenv("NVBX_FUSED_TRACE_LANES") ? atoi(getenv("NVBX_FUSED_TRACE_LANES")) : 8;       // (A/B, one frame: 4 or 8 lanes per ray)
  tr->lanes = NB == 1 ? (fused_lanes == 4 ? 4 : 8) : std::max(2, sphere_trace_lanes(c.n));
  tr->n_wg = sphere_trace_workgroups(tr->lanes, srows, scols, c.n);
  return NVBX_OK;
}
// ... its sphere tracing as a rider of the caller's launch
int nvbx_mapper::pending_color_trace_rider(void* out) {
  return color_pending.n == 1 ? trace_rider_of<1>(this, static_cast<TraceRiderT<1>*>(out)) : trace_rider_of<MAX_BATCH>(this, static_cast<TraceRiderT<MAX_BATCH>*>(out));
}
// Fused colour + TSDF launch (tsdf.hip): the held-back frame's ESDF marking pass rides in the view-marking launch of the next depth frame
// (color_launch_integrate's own part, decided before that launch) ...
void nvbx_mapper::pending_marking_args(int32_t* mark_wg, EsdfArgs* ea_out, bool single_frame) {
  *mark_wg = 0;
  EsdfArgs ea = make_esdf_args();
  const bool own = dirty_since_mark && !premark_consumed && ea.bz_hi >= ea.bz_lo && ea.bz_hi - ea.bz_lo + 1 <= 63;     // (2-D exact transform: checked by the caller)
  if (own) {
    mark_pass++; ea.mark_pass = mark_pass; unresolved_marks = true; dirty_since_mark = false; premark_consumed = !ea.self_reset;
    // one wavefront per dirty TSDF block, four per riding workgroup: 256 workgroups cover the ~300 blocks a room-sized view dirties per frame in one round; a
    // larger view (the dirty list is what the last TSDF updates had in view: the count the GPU last mirrored) gets as many as give every entry a wavefront of
    // its own, up to 1 024 -- with 256, the 2 400 entries of a 14 x 12 m hall were 2-3 dependent entries per wavefront and the marking riders the
    // longest part of the view-marking launch (18.6 us against the tiles' 16.1; tools/wg_timeline.py --scene hall: k_mark_view 17.7 -> 16.6 us, tools/mark_riders_ab.sh).
    // A BATCH keeps 256: its view-marking launch is residency-bound and more riders in front keep the tiles waiting (8 cameras: 27.0 -> 27.9 us).
    // NVBX_MARK_RIDERS=n: fixed (A/B).
    static const int fixed = getenv("NVBX_MARK_RIDERS") ? atoi(getenv("NVBX_MARK_RIDERS")) : 0;
    const int64_t n_hint = std::max<int64_t>(0, __atomic_load_n(&h_mirror[2], __ATOMIC_RELAXED));
    *mark_wg = fixed > 0 ? (fixed + 7) / 8 * 8 : (!single_frame ? 256 : (int32_t)std::min<int64_t>(1024, std::max<int64_t>(256, ((n_hint + n_hint / 4 + 3) / 4 + 7) / 8 * 8)));
  }
  *ea_out = ea;
}
// ... and its colour integration shares a launch with that frame's TSDF update: frames and scratch as for the separate launch
int nvbx_mapper::pending_color_fused_args(void* fsc_out, int* kind, int32_t* srows, int32_t* scols) {
  const ColorPending c = take_pending();      // (its frames are let go of by the caller, once the launch that reads them is enqueued: release_consumed_frames)
  *kind = c.kind;
  if (c.n > 1) { PoseSet<MAX_BATCH> ps; return pending_setup<PixRgb8, MAX_BATCH>(this, c, static_cast<FrameSetC<PixRgb8, MAX_BATCH>*>(fsc_out), &ps, srows, scols); }
  PoseSet<1> ps;
  if (c.kind == 0) return pending_setup<PixRgb8, 1>(this, c, static_cast<FrameSetC<PixRgb8, 1>*>(fsc_out), &ps, srows, scols);
  return pending_setup<PixBgra8, 1>(this, c, static_cast<FrameSetC<PixBgra8, 1>*>(fsc_out), &ps, srows, scols);
}
template <typename Pix, int NB>
static int launch_pending_integrate(nvbx_mapper* m, const nvbx_mapper::ColorPending& c) {
  FrameSetC<Pix, NB> fs; PoseSet<NB> ps; int32_t srows = 0, scols = 0;
  const int rc = pending_setup<Pix, NB>(m, c, &fs, &ps, &srows, &scols); if (rc) return rc;
  return color_launch_integrate<Pix, NB>(m, fs, srows, scols);
}
// A held-back colour frame AND the updateEsdf behind it, replayed together (a drain: synchronize, a query, any other entry point) in TWO
// launches instead of three: {sphere tracing || ESDF site marking} (both read the TSDF only), then {colour integration || distance transform}.
// The same kernels as the classic replay, the same order of dependent steps, the same map.
template <typename Pix, int NB>
static int replay_pair_t(nvbx_mapper* m, const nvbx_mapper::ColorPending& c) {
  { const int rc = color_precheck(m, c.n, c.rows, c.cols, c.T); if (rc) return rc; }
  if (m->flush_edt()) return NVBX_E_DEVICE;      // an older held-back distance transform precedes the marking pass (it reads the masks)
  FrameSetC<Pix, NB> fs; PoseSet<NB> ps; int32_t srows = 0, scols = 0;
  { const int rc = pending_setup<Pix, NB>(m, c, &fs, &ps, &srows, &scols); if (rc) return rc; }
  int32_t mark_wg = 0; EsdfArgs ea{};
  m->pending_marking_args(&mark_wg, &ea, NB == 1);        // (classic flags: the distance transform below empties the list)
  { const int rc = color_launch_trace<NB>(m, ps, c.n, srows, scols, mark_wg, ea); if (rc) return rc; }
  { const int rc = nvbx_update_esdf(m); if (rc) return rc; }      // arms the distance transform (launches the marking pass itself if it could not ride)
  return color_launch_integrate<Pix, NB>(m, fs, srows, scols, true);
}
bool nvbx_mapper::replay_pair_applies() const {
  static const int on = getenv("NVBX_REPLAY_PAIR") ? atoi(getenv("NVBX_REPLAY_PAIR")) : 1;      // (A/B: 0 = three classic launches)
  return on && color_pending.on && esdf_update_pending && p.projective_layer_type != 1 && p.esdf_mode == 0 && p.esdf_propagation == 0 && !use_side && defer_edt && !import_pending;
}
int nvbx_mapper::replay_pair() {
  const ColorPending c = take_pending(); esdf_update_pending = false;
  const int rc = c.n > 1 ? replay_pair_t<PixRgb8, MAX_BATCH>(this, c) : (c.kind == 0 ? replay_pair_t<PixRgb8, 1>(this, c) : replay_pair_t<PixBgra8, 1>(this, c));
  release_consumed_frames();
  return rc;
}
int nvbx_mapper::launch_pending_color_after_trace() {
  const ColorPending c = take_pending();
  const int rc = c.n > 1 ? launch_pending_integrate<PixRgb8, MAX_BATCH>(this, c) : (c.kind == 0 ? launch_pending_integrate<PixRgb8, 1>(this, c) : launch_pending_integrate<PixBgra8, 1>(this, c));
  release_consumed_frames();
  return rc;
}

extern "C" int nvbx_integrate_color(nvbx_mapper* m, const uint8_t* rgb_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                    const nvbx_camera* camera) {
  if (!m || !rgb_dev || !T_L_C || !camera || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_color: invalid argument (image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_color: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  { int rc = NVBX_OK; const void* im = rgb_dev; if (defer_color(m, 0, 1, &im, rows, cols, T_L_C, camera, &rc)) return rc; }
  const PixRgb8 img{rgb_dev};
  return integrate_colors<PixRgb8, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
extern "C" int nvbx_integrate_color_bgra8(nvbx_mapper* m, const uint8_t* bgra_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                          const nvbx_camera* camera) {
  if (!m || !bgra_dev || !T_L_C || !camera || !image_dims_ok(rows, cols) || ((uintptr_t)bgra_dev & 3)) { set_error("nvbx_integrate_color_bgra8: invalid argument"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_color_bgra8: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  { int rc = NVBX_OK; const void* im = bgra_dev; if (defer_color(m, 1, 1, &im, rows, cols, T_L_C, camera, &rc)) return rc; }
  const PixBgra8 img{reinterpret_cast<const uint32_t*>(bgra_dev)};
  return integrate_colors<PixBgra8, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
// integrateColor of an image in a frame of nvbx_frame_acquire / nvbx_color_image_acquire whose ownership passes to the mapper with the call
extern "C" int nvbx_integrate_color_owned(nvbx_mapper* m, void* frame, int32_t bytes_per_pixel, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera) {
  if (!m || (bytes_per_pixel != 3 && bytes_per_pixel != 4) || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_color_owned: invalid argument (3 = rgb8, 4 = bgra8)"); return NVBX_E_INVALID; }
  bool too_small = false;
  if (!frame_retain_if_frame(frame, (size_t)rows * (size_t)cols * (size_t)bytes_per_pixel, &too_small)) {       // (a reference of this call's own, for the way through)
    set_error(too_small ? "nvbx_integrate_color_owned: the frame is smaller than rows x cols pixels" : "nvbx_integrate_color_owned: not a live frame of nvbx_frame_acquire"); return NVBX_E_INVALID; }
  const int rc = bytes_per_pixel == 4 ? nvbx_integrate_color_bgra8(m, (const uint8_t*)frame, rows, cols, T_L_C, camera) : nvbx_integrate_color(m, (const uint8_t*)frame, rows, cols, T_L_C, camera);
  if (rc == NVBX_E_INVALID) { (void)nvbx_frame_release(frame); return rc; }              // (argument errors: the caller keeps the frame)
  (void)nvbx_frame_release(frame);          // the caller's reference: ownership has passed (never the last one here -- this call still holds its own)
  // a frame that was NOT held back (deferral off, an occupancy mapper) has just been read by launches enqueued on the mapper's stream: it is let go of
  // behind them, with a fence, like a held-back one -- and so is the frame of a call that failed on the device (NVBX_E_DEVICE): reader launches may
  // already be enqueued
  if (rc == NVBX_OK && m->color_pending.on && m->color_pending.frames[0] == frame) (void)nvbx_frame_release(frame);     // (held back: the mapper keeps the reference defer_color took)
  else { m->consumed_frames.push_back(frame); m->release_consumed_frames(); }
  return rc;
}
// Up to NVBX_MAX_BATCH colour frames (rgb8, same image size) in ONE launch set: see include/nvblox_hip.h
extern "C" int nvbx_integrate_color_batch(nvbx_mapper* m, int32_t n, const uint8_t* const* rgb_dev, int32_t rows, int32_t cols, const float* T_L_C,
                                          const nvbx_camera* cameras) {
  if (!m || n < 1 || n > MAX_BATCH || !rgb_dev || !T_L_C || !cameras || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_color_batch: invalid argument (1 <= n <= 8, image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  for (int c = 0; c < n; c++)
    if (!rgb_dev[c] || !nvbx_camera_matches(cameras + c, rows, cols)) { set_error("nvbx_integrate_color_batch: every camera's width/height must equal the images' cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  if (n == 1) return nvbx_integrate_color(m, rgb_dev[0], rows, cols, T_L_C, cameras);
  { int rc = NVBX_OK; if (defer_color(m, 0, n, reinterpret_cast<const void* const*>(rgb_dev), rows, cols, T_L_C, cameras, &rc)) return rc; }     // (held back: a depth batch carries it out)
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
