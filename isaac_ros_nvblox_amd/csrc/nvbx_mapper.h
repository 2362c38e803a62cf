// nvbx_mapper.h -- host-side state of one mapper (one GPU, one stream).  Host code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
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
  int32_t rec;                      // C_ESDF_UPD + 8 * (epoch & 1)
  int32_t rec_next;                 // record of the next epoch (reset by this update)
};

struct MeshRecord { int32_t x, y, z, vbase, nvert, tbase, ntri, pad; };

void set_error(const char* what, hipError_t e);
void set_error(const char* what);

}  // namespace nvbx

struct nvbx_mapper {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  nvbx_mapper_params p{};
  int64_t capacity = 0;
  nvbx::DMap d{};
  // lists (device)
  int32_t* view_list = nullptr;      // hash-entry ids of the blocks in view of the last depth frame
  int32_t* esdf_dirty = nullptr;     // slots dirtied since the last ESDF update
  int32_t* mesh_dirty = nullptr;     // 2 x capacity: slots dirtied since the last mesh update (list of parity mesh_epoch & 1 is live)
  int32_t* color_list = nullptr;     // slots updated by the last colour frame
  int32_t* export_idx = nullptr;     // int32[capacity][3] scratch for multi-GPU export of the dirty list
  int32_t* export_count = nullptr;
  // colour scratch
  float* synth = nullptr; int64_t synth_cap = 0; int32_t synth_rows = 0, synth_cols = 0;
  // mesh arena
  float* mesh_vert = nullptr; float* mesh_nrm = nullptr; uint8_t* mesh_col = nullptr; int32_t* mesh_tri = nullptr;
  nvbx::MeshRecord* mesh_rec = nullptr;
  int64_t mesh_vert_cap = 0, mesh_tri_cap = 0;
  // staging
  void* staging = nullptr; int64_t staging_bytes = 0;
  int32_t* h_counters = nullptr;     // pinned
  uint32_t frame_id = 0;
  uint32_t esdf_epoch = 0;
  uint32_t mesh_epoch = 0;
  int32_t* mesh_dirty_live() const { return mesh_dirty + (int64_t)(mesh_epoch & 1) * capacity; }
  int mesh_dirty_counter() const { return nvbx::C_MESH_DIRTY + (int)(mesh_epoch & 1); }
  uint32_t last_view_frame = 0;
  // C-ABI helpers implemented across the .hip files
  nvbx::Frame make_frame(const float T_L_C[16], const nvbx_camera* cam, int32_t rows, int32_t cols, int32_t subsample) const;
  nvbx::EsdfArgs make_esdf_args() const;
  int fetch_counters();              // D2H of all counters + stream sync
  // optional per-kernel timing (hipEvent pairs on the mapper stream), used by bench.py for the roofline line
  bool profiling = false;
  struct Span { const char* name; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t get_event();
  void span_begin(const char* name);
  void span_end();
};

// every kernel launch of the library goes through this macro (name = the kernel's name in rocprof output)
#define NVBX_LAUNCH(m, kernel, grid, block, ...)                                     \
  do {                                                                               \
    if ((m)->profiling) (m)->span_begin(#kernel);                                    \
    hipLaunchKernelGGL(kernel, grid, block, 0, (m)->stream, __VA_ARGS__);            \
    if ((m)->profiling) (m)->span_end();                                             \
  } while (0)

#define NVBX_HIP(call)                                                 \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) { nvbx::set_error(#call, e_); return NVBX_E_DEVICE; } \
  } while (0)
