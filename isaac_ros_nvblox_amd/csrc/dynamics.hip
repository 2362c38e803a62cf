// dynamics.hip -- the pieces MappingType::kDynamic adds around the hot path (nvblox_examples_bringup/config/nvblox/specializations/
// nvblox_dynamics.yaml; freespace parameters mapper_initialization.cpp:430-462): the freespace layer of the static mapper, the
// detection of dynamic depth pixels against it, and the clean-up of the dynamic mask.  The split of the depth image by that mask
// and the occupancy mapper the dynamic part goes to are the human-mapping pieces (convert.hip, tsdf.hip).
// All of it is [U] restated in oracle/nvblox_oracle.c (its freespace, dynamics-detection and component-filter restatements);
// integer timestamps, flags and masks compare bit-exactly.
#include <algorithm>
#include <cmath>
#include "nvbx_mapper.h"
#include "nvbx_mask_geom.h"

using namespace nvbx;

struct FreespaceArgs {
  int64_t now_ms;
  float max_tsdf_distance; int32_t keep_ms, free_after_ms, reset_after_ms, check_neighborhood, init_free;
};

int nvbx_mapper::ensure_freespace_pool() {
  if (d.freespace) return NVBX_OK;
  NVBX_HIP(hipMalloc(&d.freespace, (size_t)capacity * 512 * 16));
  NVBX_HIP(hipMemsetAsync(d.freespace, 0, (size_t)capacity * 512 * 16, stream));
  return NVBX_OK;
}

// One 512-thread workgroup per block of the depth frame's view list (thread = voxel, TSDF order z + 8y + 64x).  The occupied
// predicate of the 6-neighbourhood comes from a 10^3 lattice in LDS: own voxels + the touching faces of the 6 neighbour blocks.
__global__ __launch_bounds__(512) void k_update_freespace(DMap m, const int4* view_list, int32_t list_cap, int32_t cnt_idx, FreespaceArgs a) {
  __shared__ uint8_t s_occ[10 * 10 * 10];
  __shared__ uint32_t s_nb[6];
  int32_t n = m.counters[cnt_idx]; if (n > list_cap) n = list_cap;
  const int tid = threadIdx.x;
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int4 rec = view_list[i];
    const uint32_t slot = (uint32_t)rec.x;
    if (!slot_ok(slot) || !(m.slot_flags[slot] & F_TSDF)) continue;          // uniform
    __syncthreads();
    const float2 tv = m.tsdf[(size_t)slot * 512 + tid];
    const bool observed = tv.y > 0.0f;
    bool occupied = observed && tv.x < a.max_tsdf_distance;
    if (a.check_neighborhood) {
      for (int q = tid; q < 1000; q += 512) s_occ[q] = 0;
      if (tid < 6) {
        const int dx = tid == 0 ? -1 : (tid == 1 ? 1 : 0), dy = tid == 2 ? -1 : (tid == 3 ? 1 : 0), dz = tid == 4 ? -1 : (tid == 5 ? 1 : 0);
        s_nb[tid] = any_slot(m, rec.y + dx, rec.z + dy, rec.w + dz);          // TSDF pool of a slot without that layer is all-zero
      }
      __syncthreads();
      s_occ[((vx + 1) * 10 + (vy + 1)) * 10 + vz + 1] = occupied ? 1 : 0;
      if (tid < 384) {                                                          // 6 faces x 64 voxels
        const int f = tid >> 6, u = (tid >> 3) & 7, w = tid & 7;
        const uint32_t ns = s_nb[f];
        if (slot_ok(ns)) {
          int nx, ny, nz, lx, ly, lz;                                           // voxel in the neighbour block, cell in the lattice
          if (f == 0) { nx = 7; ny = u; nz = w; lx = 0; ly = u + 1; lz = w + 1; }
          else if (f == 1) { nx = 0; ny = u; nz = w; lx = 9; ly = u + 1; lz = w + 1; }
          else if (f == 2) { nx = u; ny = 7; nz = w; lx = u + 1; ly = 0; lz = w + 1; }
          else if (f == 3) { nx = u; ny = 0; nz = w; lx = u + 1; ly = 9; lz = w + 1; }
          else if (f == 4) { nx = u; ny = w; nz = 7; lx = u + 1; ly = w + 1; lz = 0; }
          else { nx = u; ny = w; nz = 0; lx = u + 1; ly = w + 1; lz = 9; }
          const float2 nv = m.tsdf[(size_t)ns * 512 + nz + 8 * ny + 64 * nx];
          s_occ[(lx * 10 + ly) * 10 + lz] = (nv.y > 0.0f && nv.x < a.max_tsdf_distance) ? 1 : 0;
        }
      }
      __syncthreads();
      if (!occupied) {
        const int c = ((vx + 1) * 10 + (vy + 1)) * 10 + vz + 1;
        occupied = s_occ[c - 100] | s_occ[c + 100] | s_occ[c - 10] | s_occ[c + 10] | s_occ[c - 1] | s_occ[c + 1];
      }
    }
    int4* fp = &m.freespace[(size_t)slot * 512 + tid];
    int4 v = *fp;
    int64_t last = (int64_t)(((u64)(uint32_t)v.y << 32) | (u64)(uint32_t)v.x);
    int32_t consec = v.z; uint32_t fl = (uint32_t)v.w;
    if (!(fl & 2u)) { fl = 2u | (a.init_free ? 1u : 0u); last = a.now_ms; consec = 0; }
    if (observed) {
      if (occupied) {
        const int64_t gap = a.now_ms - last;
        consec = gap <= (int64_t)a.keep_ms ? (int32_t)((int64_t)consec + gap) : 0;
        last = a.now_ms;
        if (consec >= a.reset_after_ms) fl &= ~1u;
      } else if (a.now_ms - last >= (int64_t)a.free_after_ms) fl |= 1u;
    }
    *fp = make_int4((int32_t)(uint32_t)((u64)last & 0xFFFFFFFFull), (int32_t)(uint32_t)((u64)last >> 32), consec, (int32_t)fl);
    if (tid == 0) atomicOr(&m.slot_flags[slot], F_FREESPACE);
  }
}

int nvbx_mapper::update_freespace() {
  if (ensure_freespace_pool()) return NVBX_E_DEVICE;
  FreespaceArgs a;
  a.now_ms = time_ms; a.max_tsdf_distance = p.max_tsdf_distance_for_occupancy_m; a.keep_ms = p.max_unobserved_to_keep_consecutive_occupancy_ms;
  a.free_after_ms = p.min_duration_since_occupied_for_freespace_ms; a.reset_after_ms = p.min_consecutive_occupancy_duration_for_reset_ms;
  a.check_neighborhood = p.check_neighborhood; a.init_free = p.initialize_to_high_confidence_freespace;
  NVBX_LAUNCH(this, k_update_freespace, dim3((unsigned)std::min<int64_t>(capacity, 1024)), dim3(512), d, (const int4*)view_list, (int32_t)capacity,
              (int32_t)(C_VIEW_COUNT + (int)(frame_id & 3)), a);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

extern "C" int nvbx_set_time_ms(nvbx_mapper* m, int64_t update_time_ms) {
  if (!m) return NVBX_E_INVALID;
  m->time_ms = update_time_ms;
  return NVBX_OK;
}

struct RtCam { float R[9], t[3], fu, fv, cu, cv; };
// [U] DynamicsDetection: the depth pixel (r, c) at depth d is dynamic iff its point lies in a high-confidence-freespace voxel
__device__ inline uint8_t pixel_is_dynamic(const DMap& m, const RtCam& g, int32_t r, int32_t c, float d, float max_d, float vs) {
  if (!(d > 0.0f) || (max_d > 0.0f && d > max_d)) return 0;
  const float rx = (((float)c + 0.5f) - g.cu) / g.fu, ry = (((float)r + 0.5f) - g.cv) / g.fv;
  float pl[3]; apply_rt(g.R, g.t, d * rx, d * ry, d, pl);
  const int32_t gx = (int32_t)floorf(pl[0] / vs), gy = (int32_t)floorf(pl[1] / vs), gz = (int32_t)floorf(pl[2] / vs);
  const uint32_t s = find_slot(m, gx >> 3, gy >> 3, gz >> 3, F_FREESPACE);
  return (slot_ok(s) && (m.freespace[(size_t)s * 512 + (gz & 7) + 8 * (gy & 7) + 64 * (gx & 7)].w & 1)) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_detect_dynamics(DMap m, RtCam g, const float* depth, int32_t rows, int32_t cols, float max_d, float vs, uint8_t* mask) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = (int32_t)(i / cols), c = (int32_t)(i - (int64_t)r * cols);
    mask[i] = pixel_is_dynamic(m, g, r, c, depth[i], max_d, vs);
  }
}
static RtCam make_rtcam(const float T_L_C[16], const nvbx_camera* camera) {
  RtCam g;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) g.R[3 * i + j] = T_L_C[4 * i + j]; g.t[i] = T_L_C[4 * i + 3]; }
  g.fu = camera->fu; g.fv = camera->fv; g.cu = camera->cu; g.cv = camera->cv;
  return g;
}
extern "C" int nvbx_detect_dynamics(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                                    float max_distance_m, uint8_t* mask_dev) {
  if (!m || !depth_dev || !T_L_C || !camera || !mask_dev || rows <= 0 || cols <= 0) { set_error("nvbx_detect_dynamics: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side_keeping_held()) return NVBX_E_DEVICE;      // (reads TSDF voxels and the freespace layer only)
  const int64_t n = (int64_t)rows * cols;
  if (!m->d.freespace) { NVBX_HIP(hipMemsetAsync(mask_dev, 0, (size_t)n, m->stream)); return NVBX_OK; }    // no freespace layer yet: nothing is dynamic
  const RtCam g = make_rtcam(T_L_C, camera);
  NVBX_LAUNCH(m, k_detect_dynamics, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), m->d, g, depth_dev, rows, cols, max_distance_m,
              m->p.voxel_size, mask_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ connected components
// [U] removeSmallConnectedComponents: 8-connected components of the non-zero pixels smaller than min_size are erased.  The output
// depends only on the component SIZES, so the labelling is free: lock-free union-find over the label image (root = smallest pixel
// index; every pixel unites with its W / NW / N / NE neighbours, roots are merged with atomicMin and retried until stable), one
// flatten pass, a wave-aggregated size count and the filter -- five launches, no iteration on the host, no synchronisation
// (round 1 iterated min-propagation to a fixed point with a host read every 8 rounds: ~10 launches + a stream drain per batch).
__device__ inline int32_t cc_find(const int32_t* label, int32_t x) {
  int32_t p = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) { x = p; p = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return x;
}
// find with path halving: every second node on the way is re-hung under its grandparent (atomicMin: labels only ever decrease towards
// the root, so concurrent writers -- other finds, a link of a node that has just stopped being a root -- can only agree on an ancestor).
// Without it a large blob's pixels form chains as long as its rows, and the occasional big mask cost 100+ us (k_cc_union max 174 us).
__device__ inline int32_t cc_find_halving(int32_t* label, int32_t x) {
  int32_t p = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    const int32_t gp = __hip_atomic_load(&label[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp == p) return p;
    atomicMin(&label[x], gp);
    x = gp; p = __hip_atomic_load(&label[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return x;
}
__device__ inline void cc_union(int32_t* label, int32_t a, int32_t b) {
  for (;;) {
    a = cc_find_halving(label, a); b = cc_find_halving(label, b);
    if (a == b) return;
    if (a > b) { const int32_t t = a; a = b; b = t; }          // a < b: b's root hangs under a
    const int32_t old = atomicMin(&label[b], a);
    if (old == b) return;                                       // b was still a root: merged
    b = old;                                                    // somebody re-rooted b meanwhile: unite a with that
  }
}
// The arrays' resting state is "every pixel its own root, both size arrays zero" -- independent of the mask, so it is established once
// per image size (k_cc_init) and RESTORED by each call's last kernel (k_cc_filter): three launches per call instead of four.
__global__ __launch_bounds__(256) void k_cc_init(int64_t n, int32_t* label, int32_t* size_a, int32_t* size_b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { label[i] = (int32_t)i; size_a[i] = 0; size_b[i] = 0; }
}
__global__ __launch_bounds__(256) void k_cc_union(const uint8_t* mask, int32_t rows, int32_t cols, int32_t* label) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!mask[i]) continue;
    const int32_t r = (int32_t)(i / cols), c = (int32_t)(i - (int64_t)r * cols);
    if (c > 0 && mask[i - 1]) cc_union(label, (int32_t)i, (int32_t)i - 1);
    if (r > 0) {
      const int64_t up = i - cols;
      if (mask[up]) cc_union(label, (int32_t)i, (int32_t)up);
      if (c > 0 && mask[up - 1]) cc_union(label, (int32_t)i, (int32_t)up - 1);
      if (c + 1 < cols && mask[up + 1]) cc_union(label, (int32_t)i, (int32_t)up + 1);
    }
  }
}
// flatten (label = root) and count: the lanes of a wavefront that share a root add their number with ONE atomic (a blob's pixels are
// neighbours in memory: 64 same-address atomics per wavefront would serialise)
__global__ __launch_bounds__(256) void k_cc_count(const uint8_t* mask, int64_t n, int32_t* label, int32_t* size) {
  const int64_t n_pad = (n + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t l = -1;
    if (i < n && mask[i]) { l = cc_find(label, (int32_t)i); label[i] = l; }
    u64 todo = __ballot(l >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int32_t ll = __shfl(l, leader);
      const u64 same = __ballot(l == ll) & todo;
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&size[ll], (int32_t)__popcll(same));
      todo &= ~same;
    }
  }
}
// removes the small blobs and restores the resting state: own label (only this thread reads label[i] here), and the size array of the
// PREVIOUS call, which nobody reads in this one (this call's is still being read by the neighbours' threads: the next call clears it)
__global__ __launch_bounds__(256) void k_cc_filter(int64_t n, int32_t* label, const int32_t* size, int32_t* size_prev, int32_t min_size, uint8_t* mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (mask[i]) { if (size[label[i]] < min_size) mask[i] = 0; label[i] = (int32_t)i; }
    size_prev[i] = 0;
  }
}
extern "C" int nvbx_remove_small_components(nvbx_mapper* m, uint8_t* mask_dev, int32_t rows, int32_t cols, int32_t min_size) {
  if (!m || !mask_dev || rows <= 0 || cols <= 0) { set_error("nvbx_remove_small_components: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  const int64_t n = (int64_t)rows * cols;
  if (3 * n + 16 > m->cc_scratch_elems) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->cc_scratch) NVBX_HIP(hipFree(m->cc_scratch));
    m->cc_scratch = nullptr; m->cc_scratch_elems = 0; m->cc_ready_n = 0;
    NVBX_HIP(hipMalloc(&m->cc_scratch, (size_t)(3 * n + 16) * 4));
    m->cc_scratch_elems = 3 * n + 16;
  }
  int32_t* label = m->cc_scratch;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  if (m->cc_ready_n != n) {            // first call / another image size: establish the resting state
    NVBX_LAUNCH(m, k_cc_init, dim3(grid), dim3(256), n, label, label + n, label + 2 * n);
    m->cc_ready_n = n; m->cc_parity = 0;
  }
  int32_t* size = label + n * (1 + m->cc_parity); int32_t* size_prev = label + n * (2 - m->cc_parity);
  m->cc_parity ^= 1;
  NVBX_LAUNCH(m, k_cc_union, dim3(grid), dim3(256), (const uint8_t*)mask_dev, rows, cols, label);
  NVBX_LAUNCH(m, k_cc_count, dim3(grid), dim3(256), (const uint8_t*)mask_dev, n, label, size);
  NVBX_LAUNCH(m, k_cc_filter, dim3(grid), dim3(256), n, label, (const int32_t*)size, size_prev, min_size, mask_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;                  // asynchronous on the mapper's stream, like the other image operations
}

// ------------------------------------------------------------------------------------------------ the front end of a dynamic-mapping frame, fused
// MultiMapper::integrateDepth(depth, T_L_C, camera) of MappingType::kDynamic (nvblox_node.cpp:1062; the node reads the mask overlay and the
// dynamic points at :1098,1108) runs, before its two integrator calls, detect dynamics -> remove small components -> split the depth image:
// as separate entry points six dependent image-sized launches and a memset (k_detect_dynamics, k_cc_union, k_cc_count, k_cc_filter, fill,
// k_mask_zmin, k_split_depth: ~30 us of an 83 us frame, each moving <= 2.8 MB).  What really orders them is three GLOBAL dependencies -- the
// union needs the neighbours' mask bits, the sizes need the finished union, the filter needs the finished sizes -- so: THREE launches, no memset.
//   A  k_dyn_detect_union   a 256-thread workgroup takes a 32 x 8 PATCH of pixels and owns its inner 30 x 7: every thread evaluates one pixel
//                           (the ring is the neighbours' neighbourhood, re-evaluated: +22 % pixels, no second dependent chain), the mask bits
//                           meet in LDS, owned pixels write their bit and unite with their W / NW / N / NE neighbours in the global label image
//                           (the same lock-free union-find); the split's nearest-depth image (atomicMin) depends on the depth image only and is
//                           filled here as well
//   B  k_cc_count           as before: flatten + wave-aggregated sizes
//   C  k_dyn_filter_split   the filter AND the split: the split of pixel i asks for the CLEANED mask at the pixel mi it lands on, i.e.
//                           mask[mi] && size[label[mi]] >= min -- evaluated on the spot, so it needs no finished filter pass; every thread also
//                           cleans its own mask bit (a reader that meets the cleaned bit instead of the raw one decides the same) and restores
//                           the resting state of the arrays the NEXT call uses (labels, sizes and nearest depths exist twice, by call parity:
//                           nothing this launch reads is reset by it)
// Results: bit-identical to the three calls (tests/test_gpu_round4.py), which stay for callers that bring their own mask.
__global__ __launch_bounds__(256) void k_dyn_init(int64_t n, int32_t* label2, int32_t* size2, uint32_t* zmin2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += (int64_t)gridDim.x * blockDim.x) {
    label2[i] = (int32_t)(i < n ? i : i - n); size2[i] = 0; zmin2[i] = 0x7F7F7F7Fu;
  }
}
// the same lock-free union-find on a patch's labels in LDS (labels only ever decrease towards the root)
__device__ inline int lds_find(int32_t* lab, int x) {
  int p = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (p != x) { x = p; p = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  return x;
}
__device__ inline void lds_union(int32_t* lab, int a, int b) {
  for (;;) {
    a = lds_find(lab, a); b = lds_find(lab, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&lab[b], a);
    if (old == b) return;
    b = old;
  }
}
constexpr int DYN_PW = 32, DYN_PH = 8, DYN_OW = DYN_PW - 2, DYN_OH = DYN_PH - 1;       // patch; owned inner region (left + right column and top row are the ring)
__global__ __launch_bounds__(256) void k_dyn_detect_union(DMap m, RtCam g, MaskGeom mg, const float* depth, int32_t rows, int32_t cols, float max_d, float vs,
                                                         int32_t has_freespace, int32_t tiles_x, uint8_t* mask, int32_t* label, uint32_t* zmin) {
  __shared__ uint8_t s_mask[DYN_PH][DYN_PW];
  __shared__ int32_t s_lab[DYN_PH * DYN_PW];
  const int px = threadIdx.x & (DYN_PW - 1), py = threadIdx.x / DYN_PW;
  const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
  const int32_t r = ty * DYN_OH - 1 + py, c = tx * DYN_OW - 1 + px;
  const bool in_img = r >= 0 && c >= 0 && r < rows && c < cols;
  const int32_t i = in_img ? r * cols + c : 0;
  const float d = in_img ? depth[i] : 0.0f;
  const bool owned = in_img && px >= 1 && px <= DYN_OW && py >= 1;
  // the split's nearest depth per mask pixel: needs the depth image only (positive floats order like their bit patterns)
  if (owned && d > 0.0f) { float z; const int32_t mi = mask_pixel(mg, r, c, d, &z); if (mi >= 0) atomicMin(&zmin[mi], __float_as_uint(z)); }
  const uint8_t mk = (in_img && has_freespace) ? pixel_is_dynamic(m, g, r, c, d, max_d, vs) : (uint8_t)0;
  s_mask[py][px] = mk;
  s_lab[threadIdx.x] = (int32_t)threadIdx.x;
  const bool own_mk = owned && mk;
  if (!__syncthreads_or(own_mk ? 1 : 0)) { if (owned) mask[i] = mk; return; }       // (most patches: nothing dynamic in them)
  if (owned) mask[i] = mk;
  // Components of the patch's OWNED pixels in LDS first (the same union-find on 256 labels), so that the global label image sees one link per
  // pixel towards its patch-local root instead of up to four towards its neighbours: a large blob used to form label chains as long as its
  // rows, and every hop of a find is a dependent HBM round trip (k_dyn_detect_union 5 us without dynamics, 115 us on the largest mask).
  const int t = (int)threadIdx.x;
  if (own_mk) {
    if (px - 1 >= 1 && s_mask[py][px - 1]) lds_union(s_lab, t, t - 1);
    if (py - 1 >= 1) {
      if (s_mask[py - 1][px]) lds_union(s_lab, t, t - DYN_PW);
      if (px - 1 >= 1 && s_mask[py - 1][px - 1]) lds_union(s_lab, t, t - DYN_PW - 1);
      if (px + 1 <= DYN_OW && s_mask[py - 1][px + 1]) lds_union(s_lab, t, t - DYN_PW + 1);
    }
  }
  __syncthreads();
  if (!own_mk) return;
  const int lr = lds_find(s_lab, t);                          // patch-local root: the component's first owned pixel in row-major order
  const int32_t gr = i - (py - lr / DYN_PW) * cols - (px - (lr & (DYN_PW - 1)));       // its pixel index
  // pixels another patch's links may name (its W / NW / N / NE neighbours lie in my last row, my first and my last column) can have been
  // re-hung already: a proper union; nobody holds a pointer to the others before this store
  const bool shared = py == DYN_PH - 1 || px == 1 || px == DYN_OW;
  if (lr != t) { if (shared) cc_union(label, i, gr); else atomicMin(&label[i], gr); }
  // links that leave the owned region (the ring: left / right column, top row of the patch)
  if (px == 1 && s_mask[py][px - 1]) cc_union(label, i, i - 1);
  if (py == 1) {
    if (s_mask[py - 1][px]) cc_union(label, i, i - cols);
    if (s_mask[py - 1][px - 1]) cc_union(label, i, i - cols - 1);
    if (s_mask[py - 1][px + 1]) cc_union(label, i, i - cols + 1);
  } else {
    if (px == 1 && s_mask[py - 1][px - 1]) cc_union(label, i, i - cols - 1);
    if (px == DYN_OW && s_mask[py - 1][px + 1]) cc_union(label, i, i - cols + 1);
  }
}
__global__ __launch_bounds__(256) void k_dyn_filter_split(MaskGeom g, const float* depth, uint8_t* mask, const int32_t* label, const int32_t* size, int32_t min_size, float thr,
                                                         const uint32_t* zmin, float* unmasked, float* masked, uint8_t* overlay,
                                                         int32_t* label_next, int32_t* size_next, uint32_t* zmin_next) {
  const int64_t n = (int64_t)g.rows * g.cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = depth[i];
    bool is_masked = false;
    if (d > 0.0f) {
      float z; const int32_t mi = mask_pixel(g, (int32_t)(i / g.cols), (int32_t)(i % g.cols), d, &z);
      if (mi >= 0 && mask[mi] != 0 && (min_size <= 0 || size[label[mi]] >= min_size) && z <= __uint_as_float(zmin[mi]) + thr) is_masked = true;
    }
    if (min_size > 0 && mask[i] && size[label[i]] < min_size) mask[i] = 0;          // this pixel's own bit, cleaned
    unmasked[i] = (d > 0.0f && is_masked) ? NVBX_MASKED_DEPTH_INVALID : d;
    masked[i] = (d > 0.0f && is_masked) ? d : NVBX_MASKED_DEPTH_INVALID;
    if (overlay) {
      float gv = d > 0.0f ? d * (255.0f / 5.0f) : 0.0f; if (gv > 255.0f) gv = 255.0f;
      const uint8_t q = (uint8_t)gv;
      overlay[3 * i] = is_masked ? 255 : q; overlay[3 * i + 1] = q; overlay[3 * i + 2] = q;
    }
    label_next[i] = (int32_t)i; size_next[i] = 0; zmin_next[i] = 0x7F7F7F7Fu;       // resting state of the other parity (left behind by the call before this one)
  }
}
extern "C" int nvbx_dynamic_depth_split(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                                        float max_distance_m, int32_t min_component_size, float occlusion_threshold_m,
                                        uint8_t* mask_dev, float* depth_unmasked_dev, float* depth_masked_dev, uint8_t* overlay_rgb_dev) {
  if (!m || !depth_dev || !T_L_C || !camera || !mask_dev || !depth_unmasked_dev || !depth_masked_dev || !image_dims_ok(rows, cols)) {
    set_error("nvbx_dynamic_depth_split: invalid argument (image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side_keeping_held()) return NVBX_E_DEVICE;      // (reads TSDF voxels and the freespace layer only: held-back work stays held back, as for nvbx_detect_dynamics)
  const int64_t n = (int64_t)rows * cols;
  if (6 * n + 16 > m->dyn_scratch_elems) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->dyn_scratch) NVBX_HIP(hipFree(m->dyn_scratch));
    m->dyn_scratch = nullptr; m->dyn_scratch_elems = 0; m->dyn_ready_n = 0;
    NVBX_HIP(hipMalloc(&m->dyn_scratch, (size_t)(6 * n + 16) * 4));
    m->dyn_scratch_elems = 6 * n + 16;
  }
  int32_t* label2 = m->dyn_scratch; int32_t* size2 = label2 + 2 * n; uint32_t* zmin2 = reinterpret_cast<uint32_t*>(size2 + 2 * n);
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  if (m->dyn_ready_n != n) {            // first call / another image size: resting state of both parities
    NVBX_LAUNCH(m, k_dyn_init, dim3(grid), dim3(256), n, label2, size2, zmin2);
    m->dyn_ready_n = n; m->dyn_parity = 0;
  }
  const int par = m->dyn_parity; m->dyn_parity ^= 1;
  int32_t* label = label2 + (size_t)par * n; int32_t* size = size2 + (size_t)par * n; uint32_t* zmin = zmin2 + (size_t)par * n;
  const RtCam g = make_rtcam(T_L_C, camera);
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const MaskGeom mg = make_mask_geom(I, camera, camera, rows, cols, rows, cols);       // (the mask is the depth camera's own: MultiMapper::integrateDepth, dynamic mode)
  const int32_t tiles_x = (cols + DYN_OW - 1) / DYN_OW, tiles_y = (rows + DYN_OH - 1) / DYN_OH;
  NVBX_LAUNCH(m, k_dyn_detect_union, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), m->d, g, mg, depth_dev, rows, cols, max_distance_m, m->p.voxel_size,
              (int32_t)(m->d.freespace ? 1 : 0), tiles_x, mask_dev, label, zmin);
  if (min_component_size > 0) NVBX_LAUNCH(m, k_cc_count, dim3(grid), dim3(256), (const uint8_t*)mask_dev, n, label, size);
  NVBX_LAUNCH(m, k_dyn_filter_split, dim3(grid), dim3(256), mg, depth_dev, mask_dev, (const int32_t*)label, (const int32_t*)size, min_component_size, occlusion_threshold_m,
              (const uint32_t*)zmin, depth_unmasked_dev, depth_masked_dev, overlay_rgb_dev, label2 + (size_t)(1 - par) * n, size2 + (size_t)(1 - par) * n, zmin2 + (size_t)(1 - par) * n);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
