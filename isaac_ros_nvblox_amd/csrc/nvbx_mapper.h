// nvbx_mapper.h -- host-side state of one mapper (one GPU, one stream).  Host code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cmath>
#include <vector>
#include "../../include/nvblox_hip.h"
#include "nvbx_internal.h"

namespace nvbx {

struct EsdfArgs {
  int32_t kz_min, kz_max, kz_out;   // global voxel z range of the TSDF band, output plane
  int32_t bz_lo, bz_hi, bz_out, vz_out;
  int32_t ri;                       // integer search radius (voxels)
  int32_t rb;                       // ceil(ri / 8) blocks
  float max_sq, site_dist_m, min_weight, voxel_size;
  int32_t site_rule;
  uint32_t epoch;
  uint32_t mark_pass;               // stamp of this marking pass (column de-duplication within one pass)
  int32_t rec;                      // C_ESDF_UPD + 8 * (epoch & 1)
  int32_t rec_next;                 // record of the next epoch (reset by this update)
  // Who empties the ESDF-dirty list a marking pass has consumed.  Classic order (marking -> view marking of the next frame / EDT -> TSDF
  // update): the next k_mark_view or the EDT does (the consumed list still names "the blocks dirtied since the last updateEsdf" for
  // nvbx_esdf_dirty_list until then).  Pipelined order (colour deferral, DESIGN.md 2.8: view marking(i+1) || sphere tracing(i) -> colour(i) +
  // marking -> TSDF update(i+1)): nothing runs between the marking pass and the next appends, so the pass empties the list itself --
  // its last worker, counted in C_MARK_DONE (self_reset) -- and the EDT of that update, which then runs AFTER those appends, must not (keep_list).
  int32_t self_reset, keep_list;
  // [U] ground-plane-relative slice (nvbx_mapper_params::esdf_use_ground_plane): the z band of column (x, y) is [h + above, h + above + thick],
  // h = height of the plane {n, d} at the column's centre (esdf_plane_height); 0 = the fixed band kz_min .. kz_max above
  int32_t plane_on; float pl[4], above, thick;
};

struct MeshRecord { int32_t x, y, z, vbase, nvert, tbase, ntri, pad; };

float log_odds(float p);
// frames.hip: library-owned, reference-counted device frames (ownership transfer of input images)
bool frame_retain_if_frame(const void* p, size_t need_bytes, bool* too_small);
void frame_release_fenced(void* p, const volatile int32_t* progress, int32_t seq, const volatile int32_t* reports_enqueued, hipStream_t reader, const void* owner);
void frames_register_stream(int device, hipStream_t stream);
void frames_forget_owner(const void* owner, int device, hipStream_t stream);
void set_error(const char* what, hipError_t e);
void set_error(const char* what);

}  // namespace nvbx

// the camera model must describe the image that is handed over with it: the kernels bound projections by (width, height) and
// address pixels by (rows, cols)
static inline bool nvbx_camera_matches(const nvbx_camera* c, int32_t rows, int32_t cols) {
  return c && c->width == cols && c->height == rows && c->fu > 0.0f && c->fv > 0.0f;
}

// Block indices are 21 bits per axis in the hash key (pack_key): a sensor pose is accepted only if it is finite and everything
// within `reach` metres of it stays inside [-2^20, 2^20) blocks (419 km at 0.05 m voxels) -- beyond that keys would alias.
static inline bool nvbx_pose_in_range(const float T[16], float block_size, float reach) {
  for (int i = 0; i < 16; i++) if (!std::isfinite(T[i])) return false;
  const float lim = ((float)(1 << 20) - 2.0f) * block_size - reach;
  return std::fabs(T[3]) < lim && std::fabs(T[7]) < lim && std::fabs(T[11]) < lim;
}
static inline bool nvbx_index_in_range(int32_t x, int32_t y, int32_t z) {
  const int32_t L = 1 << 20;
  return x >= -L && x < L && y >= -L && y < L && z >= -L && z < L;
}

struct nvbx_mapper {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false, stream_registered = false;
  // Side stream: the ESDF update (k_esdf_mark + k_esdf_edt) is independent of colour integration, so it
  // runs on its own stream behind an event recorded after the last non-colour operation and overlaps integrateColor.
  // Every other entry point joins the side stream first, so the caller still sees single-stream ordering.
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  hipEvent_t ev_order = nullptr;     // nvbx_mapper_wait_for: this mapper's stream as the producer
  bool use_side = false;        // NVBX_SIDE_STREAM=0 disables
  bool side_pending = false;    // ESDF work enqueued on `side` that `stream` has not waited for yet
  bool main_dirty = true;       // non-colour work enqueued on `stream` since ev_main was recorded
  int join_side();              // make `stream` wait for the side stream (no host sync)
  int mark_main();              // record ev_main on `stream` (call right after a non-colour operation)
  nvbx_mapper_params p{};
  int64_t capacity = 0;
  // Pool growth (the reference allocates blocks on demand; here the pools double before they run out): kernels mirror the number of
  // free slots into pinned host memory (DMap::host_mirror), every integrate call looks at it first and, below half, doubles every
  // capacity-sized array + rebuilds the hash on the device (grow_map, maintenance.hip).  max_capacity == capacity: fixed pools.
  int64_t max_capacity = 0;
  int32_t* h_mirror = nullptr;       // host view of DMap::host_mirror
  int maybe_grow(int64_t extra_blocks_wanted = 0);
  int grow_map(int64_t new_capacity);
  int64_t growths = 0;
  nvbx::DMap d{};
  // lists (device)
  int32_t* view_list = nullptr;      // int4 {slot, x, y, z} of the blocks in view of the last depth frame
  // (the ESDF-dirty / mesh-dirty / colour work lists are sharded and live in d.lists / d.shc, nvbx_internal.h)
  int32_t* export_idx = nullptr;     // int32[capacity][3] scratch for multi-GPU export of the dirty list
  int32_t* export_count = nullptr;
  int32_t* cleared_idx = nullptr;    // int32[capacity][3]: Index3D of projective-layer blocks deallocated since nvbx_take_cleared_blocks
  // LiDAR beam direction tables (float2 {sin, cos}: rows elevations then cols azimuths), rebuilt when the model changes
  void* lidar_tab = nullptr; size_t lidar_tab_cap = 0; nvbx_lidar lidar_cached{}; std::vector<float> lidar_host;
  // mask splitting scratch: nearest depth (in the mask camera) that landed on each mask pixel
  uint32_t* mask_zmin = nullptr; int64_t mask_zmin_cap = 0;
  // depth preprocessing scratch (dilated depth image)
  float* depth_pre = nullptr; int64_t depth_pre_cap = 0;
  // colour scratch
  float* synth = nullptr; int64_t synth_cap = 0; int32_t synth_rows = 0, synth_cols = 0, synth_last = 0;
  // mesh arena
  float* mesh_vert = nullptr; float* mesh_nrm = nullptr; uint8_t* mesh_col = nullptr; int32_t* mesh_tri = nullptr;
  nvbx::MeshRecord* mesh_rec = nullptr;
  int64_t mesh_vert_cap = 0, mesh_tri_cap = 0;
  // staging
  void* staging = nullptr; int64_t staging_bytes = 0;
  // nvbx_esdf_slice_to_host: pinned, device-mapped [16-int header][image] that k_esdf_slice_rows writes directly (one wait per host slice)
  float* slice_pinned = nullptr; float* slice_pinned_dev = nullptr; int64_t slice_pinned_elems = 0;
  // every kernel launch bumps enqueue_seq (NVBX_LAUNCH*); nvbx_esdf_slice_size remembers it: a slice_to_image right behind it re-uses the counters
  uint64_t enqueue_seq = 1, slice_size_seq = 0;
  int32_t* h_counters = nullptr;     // pinned
  uint32_t frame_id = 0;
  uint32_t esdf_epoch = 0;
  uint32_t mark_pass = 0;            // ESDF marking passes launched so far
  // ESDF marking state: `dirty_since_mark` = TSDF changed since the last marking pass; `premark_consumed` = the dirty list
  // was processed by a pass that no EDT followed yet, so it must be emptied before anything is appended to it
  bool dirty_since_mark = false, premark_consumed = false;
  // `unresolved_marks` = a marking pass ran that no distance transform has followed yet (its columns are only PENDING):
  // an operation that deallocates blocks takes such passes back first (undo_marks, esdf.hip), so that the ESDF update that
  // eventually runs sees exactly the dirty set the reference semantics define at that moment.  `pass_at_last_edt` =
  // mark_pass when the last distance transform was enqueued (passes above it are the unresolved ones).
  bool unresolved_marks = false; uint32_t pass_at_last_edt = 0;
  int undo_marks();
  // nvbx_tsdf_zero_crossings: the sorted result of the last call, valid until any other entry point runs (join_side) -- serves the
  // "count, then data" pair of calls with one launch, one download and one sort
  bool zc_valid = false; float zc_min = 0.0f, zc_max = 0.0f; std::vector<float> zc_points;
  // dynamic mapping (dynamics.hip)
  int64_t time_ms = 0;               // update_time_ms of the next integrateDepth
  int ensure_freespace_pool();
  int update_freespace();            // after the TSDF update of a depth frame (projective_layer_type 2)
  int32_t* apply_postab = nullptr; int64_t apply_postab_cap = 0;      // nvbx_apply_measurements: per slot, the record position of each rank (+1)
  int32_t* cc_scratch = nullptr; int64_t cc_scratch_elems = 0;      // connected components: label[n], size[2][n]
  int64_t cc_ready_n = 0; int cc_parity = 0;                         // image size the arrays are initialised for; which size array the next call uses
  int32_t* dyn_scratch = nullptr; int64_t dyn_scratch_elems = 0;    // nvbx_dynamic_depth_split: label[2][n], size[2][n], nearest depth[2][n] (by call parity)
  int64_t dyn_ready_n = 0; int dyn_parity = 0;
  // EsdfMode::k3D (esdf3d.hip)
  int update_esdf_3d();
  void* esdf3_scratch = nullptr; int64_t esdf3_scratch_bytes = 0; int64_t esdf3_blocks_marked = 0, esdf3_window_voxels = 0;
  // held-back EDT of the last updateEsdf (NVBX_DEFER_EDT=0 disables): see nvbx_update_esdf
  bool defer_edt = true, edt_pending = false; nvbx::EsdfArgs edt_args{};
  int flush_edt();
  // held-back union step of the multi-GPU exchange (nvbx_mark_esdf_dirty_gathered_deferred): rides in the next integrateColor
  // launch beside the marking of the mapper's own dirty blocks; every other entry point launches it first (flush_import)
  bool import_pending = false; const int32_t* import_ptr = nullptr; int32_t import_world = 0, import_rank = 0; int64_t import_max = 0;
  int flush_import();
  // Colour deferral (nvbx_mapper_set_color_deferral): integrateColor of a single frame is HELD BACK -- arguments remembered,
  // nothing launched -- and so is an updateEsdf that follows it.  The next single-camera integrateDepth carries them out in pipelined
  // order: view marking of the new depth frame || sphere tracing of the held-back colour frame (one launch), colour integration + ESDF
  // marking, TSDF update of the new frame -- three launches per frame instead of four.  Every other entry point first replays the held-back
  // calls as they are (join_side -> replay_deferred), so the API observes call order.  Contract: the colour image must stay valid and
  // unchanged until the next call into the mapper has returned.
  // (n = 1: integrateColor, kind 0 = rgb8 / 1 = bgra8; n > 1: nvbx_integrate_color_batch, rgb8 -- carried out by a depth BATCH in pipelined order)
  // (frames[i] != nullptr: imgs[i] lives in a library-owned frame this mapper has RETAINED -- the caller's own frame of nvbx_frame_acquire, or the
  //  frame the staged form copied a raw pointer's image into; let go of, with a fence, once the launches that read it are enqueued: frames.hip)
  struct ColorPending { bool on = false; int kind = 0; int32_t n = 1; const void* imgs[nvbx::MAX_BATCH] = {}; void* frames[nvbx::MAX_BATCH] = {}; int32_t rows = 0, cols = 0; float T[16 * nvbx::MAX_BATCH]; nvbx_camera cams[nvbx::MAX_BATCH]; };
  bool color_deferral = true;        // the switch (default: on, staged -- nvbx_mapper_create; NVBX_COLOR_DEFERRAL=0 in the environment: off)
  // nvbx_mapper_set_color_deferral(m, 2): a held-back frame is COPIED into mapper-owned staging memory when it is held back (one asynchronous
  // device-to-device copy on the mapper's stream per frame) -- the caller may recycle or overwrite its image as soon as integrateColor has
  // returned, as without deferral.  One buffer per camera of a batch suffices: the copy of the next frame is stream-ordered behind the launches
  // that read the previous one.
  bool color_staging = true;
  ColorPending color_pending;        // the held-back integrateColor
  // frames of held-back colour images (frames.hip).  take_pending: the held-back call is being carried out -- its frames move to `consumed_frames`;
  // release_consumed_frames (after the launches that read them are enqueued, or abandoned): refs dropped with the fence {h_mirror[4] >= seq}.
  // h_mirror[4] is written by the next view-marking launch as its first action (TraceRiderT::fence_report = color_reads_enqueued at that time).
  std::vector<void*> consumed_frames;
  int32_t color_reads_enqueued = 0;      // fence sequence: bumped once per release_consumed_frames
  int32_t fence_reports_enqueued = 0;    // the largest fence_report any ENQUEUED launch carries
  ColorPending take_pending() { ColorPending c = color_pending; color_pending.on = false; for (int i = 0; i < nvbx::MAX_BATCH; i++) { if (c.frames[i]) consumed_frames.push_back(c.frames[i]); color_pending.frames[i] = nullptr; } return c; }
  void release_consumed_frames() {
    if (consumed_frames.empty()) return;
    const int32_t seq = ++color_reads_enqueued;
    for (void* f : consumed_frames) nvbx::frame_release_fenced(f, h_mirror + 4, seq, &fence_reports_enqueued, stream, this);
    consumed_frames.clear();
  }
  int32_t next_fence_report() { __atomic_store_n(&fence_reports_enqueued, color_reads_enqueued, __ATOMIC_RELEASE); return color_reads_enqueued; }
  int64_t inv_i8 = 0;                // (-DNVBX_CHECK_INVARIANTS) colour-reading launches enqueued on a library frame that nobody holds
  bool esdf_update_pending = false;  // an updateEsdf called while a colour frame was held back
  bool replaying = false;            // inside replay_deferred: the calls run as usual
  bool pipelined_order = false;      // inside the pipelined integrateDepth: marking passes empty their list, EDTs keep it (EsdfArgs)
  int replay_deferred();
  // join_side for an entry point that neither reads nor writes what the held-back work touches (colour layer, ESDF layer, site masks, the
  // ESDF-dirty list): the held-back distance transform, union step, colour frame and ESDF update stay held back (nvbx_detect_dynamics: TSDF +
  // freespace reads only -- the dynamic-mapping frame starts with it, and flushing there would cost the pipeline every frame)
  int join_side_keeping_held();
  // a held-back updateEsdf with NO colour frame in front of it (colour deferral on, the caller integrates no colour: depth-only hosts, occupancy
  // mappers) that the next camera launch can carry in two-launch order: marking pass in the view-marking launch, distance transform in the
  // TSDF-update launch
  bool esdf_only_carry() const { return esdf_update_pending && !color_pending.on && p.esdf_mode == 0 && p.esdf_propagation == 0 && !import_pending && !use_side && defer_edt; }
  int pending_color_trace_rider(void* trace_rider_out);   // color.hip: set the held-back frame(s) up; the sphere tracing as a nvbx::TraceRiderT<1> (one frame) / <MAX_BATCH> (a batch)
  int launch_pending_color_after_trace();
  bool replay_pair_applies() const;  // color.hip: a held-back colour frame + updateEsdf can be replayed in two launches (replay_pair)
  int replay_pair();
// -- fused colour + TSDF launch of the pipelined order (two launches per frame)
  int4* color_cand = nullptr;        // [2][fuse_cap] candidate records {slot, block index} of the held-back colour frame (parity cand_parity)
  int64_t fuse_cap = 0;
  int cand_parity = 0;
  bool lidar_integrated = false;     // a LiDAR scan has been integrated since the last clear: blocks may be F_BAND_STALE -> no fused launches
  int ensure_fuse_buffers();         // tsdf.hip
  // color.hip: the marking pass that rides in the view-marking launch (0 workgroups: none), and the held-back colour frame's set-up for the
  // fused launch (FrameSetC<Pix, 1>: rgb8 / bgra8 share one layout; FrameSetC<PixRgb8, MAX_BATCH> for a held-back batch)
  void pending_marking_args(int32_t* mark_wg, nvbx::EsdfArgs* ea_out, bool single_frame);
  int pending_color_fused_args(void* fsc_out, int* kind, int32_t* srows, int32_t* scols);
  void* table_spare = nullptr; void* table_dirty = nullptr; uint32_t table_mask_extra = 0xFFFFFFFFu;       // decay's rotating hash tables: the all-empty one k_decay builds the next table in, and the one it empties for the call after (maintenance.hip, round 6)
  uint8_t* view_class = nullptr; int64_t view_class_cap = 0;        // LiDAR: per view record, 1 = updated by the beam-centric launch (tsdf.hip k_lidar_sparse)
  // LiDAR view calculation over a dense grid (tsdf.hip k_mark_view_grid): one byte per block of the box around the sensor (cell-major, 64 B per
  // 4 x 4 x 4 cell) + one byte per cell; all-zero between scans (k_scan_view_grid puts back what the scan set).  `view_grid_dirty`: a scan's
  // launches were not all enqueued (an error return in between) -- the next scan clears the arrays first.
  uint8_t* view_grid_fine = nullptr; int64_t view_grid_cells_cap = 0; bool view_grid_dirty = false;
  int32_t* view_export = nullptr; int64_t view_export_cap = 0;      // nvbx_set_view_export
  int reset_consumed_list();         // empty a consumed dirty list (tiny launch; rare paths only)
  int begin_dirtying() { const int rc = reset_consumed_list(); dirty_since_mark = true; return rc; }
  uint32_t mesh_epoch = 0;
  int mesh_list_live() const { return nvbx::S_LIST_MESH_DIRTY + (int)(mesh_epoch & 1); }   // mesh-dirty list being filled
  int32_t* h_shc = nullptr;          // pinned mirror of d.shc
  int64_t shc_sum(int id, int field) const { int64_t t = 0; for (int s = 0; s < nvbx::NSH; s++) t += h_shc[(id * nvbx::NSH + s) * nvbx::SH_STRIDE + field]; return t; }
  uint32_t last_view_frame = 0;
  // frame stamp of the last CAMERA depth frame: decayTsdfExcludeLastView<Camera> spares the camera's view only -- a LiDAR scan in
  // between must not take its place (nvblox_node.cpp:931-936: 'lidar views are not excluded')
  uint32_t last_camera_view_frame = 0;
  uint32_t last_camera_view_mask = 1;  // camera bit (Entry::stamp) of the last camera of that frame / batch
  int32_t last_view_batch = 1;         // frames in the last depth launch set
  // C-ABI helpers implemented across the .hip files
  nvbx::Frame make_frame(const float T_L_C[16], const nvbx_camera* cam, int32_t rows, int32_t cols, int32_t subsample) const;
  nvbx::EsdfArgs make_esdf_args() const;
  int fetch_counters();              // D2H of all counters + stream sync
  // wait until the mapper's stream has done everything enqueued so far (queries: the slice for a host, counters, nvbx_synchronize)
  int wait_stream();
  // optional per-kernel timing (hipEvent pairs on the mapper stream), used by bench.py for the roofline line
  bool profiling = false;
  struct Span { const char* name; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t get_event();
  void span_begin(const char* name, hipStream_t s);
  void span_end(hipStream_t s);
};

// every kernel launch of the library goes through this macro (name = the kernel's name in rocprof output)
#define NVBX_LAUNCH_ON(m, s, kernel, grid, block, ...)                               \
  do {                                                                               \
    if ((m)->profiling) (m)->span_begin(#kernel, (s));                               \
    (m)->enqueue_seq++;                                                              \
    hipLaunchKernelGGL(kernel, grid, block, 0, (s), __VA_ARGS__);                    \
    if ((m)->profiling) (m)->span_end((s));                                          \
  } while (0)
#define NVBX_LAUNCH(m, kernel, grid, block, ...) NVBX_LAUNCH_ON(m, (m)->stream, kernel, grid, block, __VA_ARGS__)
// (... with `smem` bytes of dynamic LDS per workgroup)
#define NVBX_LAUNCH_SMEM(m, kernel, grid, block, smem, ...)                          \
  do {                                                                               \
    if ((m)->profiling) (m)->span_begin(#kernel, (m)->stream);                       \
    (m)->enqueue_seq++;                                                              \
    hipLaunchKernelGGL(kernel, grid, block, (smem), (m)->stream, __VA_ARGS__);       \
    if ((m)->profiling) (m)->span_end((m)->stream);                                  \
  } while (0)

#define NVBX_HIP(call)                                                 \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) { nvbx::set_error(#call, e_); return NVBX_E_DEVICE; } \
  } while (0)
