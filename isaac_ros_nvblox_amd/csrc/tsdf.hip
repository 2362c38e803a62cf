// tsdf.hip -- MultiMapper::integrateDepth on MI355X: block marking (view calculation) + projective TSDF update, for
// the pinhole camera and for the spinning LiDAR (range image).
//
// Two launches per depth frame, no host round trip in between:
//   k_mark_view      one wavefront per 8x8 tile of the sub-sampled ray grid.  Phase 1: each lane walks its ray through
//                    the block grid (Amanatides-Woo) and drops the block keys into a 4-8 KiB LDS set (rays of one tile
//                    share almost all their blocks) -- no HBM access inside the walk.  Phase 2 (flush): the set is
//                    compacted (ballot + popcount) and ONE key per lane goes to HBM: CAS insert-if-absent into the hash
//                    (device-side allocation from the slot stack), per-entry frame stamp, and a wave-aggregated append
//                    of {slot, Index3D} to the frame's view list (exactly once per block and frame).  A camera tile
//                    (< 100 blocks) flushes once; long LiDAR rays flush whenever the set is half full.
//   k_integrate_tsdf one 512-thread workgroup (8 wave64) per 8^3 block, grid-striding over the device-resident view
//                    list of {slot, Index3D} records; lane = voxel in z + 8y + 64x order, so every wave reads/writes
//                    512 contiguous bytes.
// Both kernels are templated on the depth source (f32 metres / u16 millimetres) and on the sensor model.
// Reference semantics restated: [U] ViewCalculator::getBlocksInImageViewRaycast and ProjectiveTsdfIntegrator
// (call sites nvblox_ros/src/lib/nvblox_node.cpp:1062 camera, :1382-1384 LiDAR; knobs mapper_initialization.cpp:264-358).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#define NVBX_WGT_HERE
#include "nvbx_mapper.h"
#include "nvbx_lidar_math.h"
#include "nvbx_esdf_edt.h"
#include "nvbx_sphere_trace.h"
#include "nvbx_esdf_mark.h"
#include "nvbx_color_worker.h"

using namespace nvbx;

// Per-workgroup time stamps of the two camera launches (tools/wg_timeline.py builds a variant of the library with -DNVBX_WG_TIMES; the product
// build compiles NVBX_T to nothing).  s_memrealtime: a constant 100 MHz clock shared by all CUs; slot i of workgroup blockIdx.x of kernel k.
#ifdef NVBX_WG_TIMES
namespace nvbx { __device__ unsigned long long* g_wgt = nullptr; }
constexpr int WGT_MAX_WG = 8192, WGT_SLOTS = 8;
#define NVBX_T(k, i) NVBX_TV(k, i, wall_clock64())
static unsigned long long* g_wgt_host = nullptr;
extern "C" int nvbx_debug_wg_times(unsigned long long* out_host, int64_t n_words) {
  const size_t total = (size_t)2 * WGT_MAX_WG * WGT_SLOTS;
  if (!g_wgt_host) {
    if (hipMalloc(&g_wgt_host, total * 8) != hipSuccess) return -1;
    (void)hipMemset(g_wgt_host, 0, total * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgt), &g_wgt_host, sizeof(g_wgt_host));
  }
  if (out_host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out_host, g_wgt_host, std::min<size_t>(total, (size_t)n_words) * 8, hipMemcpyDeviceToHost);
    (void)hipMemset(g_wgt_host, 0, total * 8);
  }
  return (int)WGT_MAX_WG;
}
#else
#define NVBX_T(k, i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ sensor models
// Camera(fu, fv, cu, cv, w, h): conversions/image_conversions.cpp:27-32.  Everything it needs is in Frame.
struct CameraSensor {
  static constexpr bool kLongRays = false;
#ifndef NVBX_CAM_TR
#define NVBX_CAM_TR 8
#define NVBX_CAM_TC 8
#define NVBX_CAM_SEG 1
#endif
#ifndef NVBX_CAM_GR
#define NVBX_CAM_GR 2
#define NVBX_CAM_GC 2
#endif
  static constexpr int kTileRows = NVBX_CAM_TR, kTileCols = NVBX_CAM_TC;     // rays per wavefront: one tile of the ray grid (tools/lidar_tile_sweep.sh cam)
  static constexpr int kSetSize = 512, kFlushRounds = 2; // a tile crosses < 100 blocks: 4 KiB set, 128 keys per flush pass
  static constexpr int kSegments = NVBX_CAM_SEG;         // lanes per ray
  static constexpr int kProbeDepth = 2;                  // hash probe positions fetched up front per key in a flush
  static constexpr bool kRiders = true;                  // workers of other passes may ride in the view-marking launch (256 threads each: a riding worker uses four wavefronts as it likes)
  static constexpr int kThreads = 64 * NVBX_CAM_GR * NVBX_CAM_GC;      // one wavefront per tile: the tiles of a group share the workgroup's key set
  // Tiles per workgroup: kGroupRows x kGroupCols NEIGHBOURING tiles share ONE LDS key set.  Every ray starts in the camera's block and
  // the rays of neighbouring tiles run through the same blocks for their first metres, so with one tile per workgroup the block at the
  // origin had its stamp claimed by ALL 336 tiles of a 640x480 frame at the same moment -- returning atomics on one address serialise
  // at ~12 ns each in the memory-side atomic unit (tools/micro/atomic_scope_bench.hip: 336 of them = 4.0 us for the last; whatever the
  // scope, there are no XCD-local atomics) -- and the tiles' flush took 4.2 of their 10.7 us (tools/wg_timeline.py).  Four tiles per set:
  // a quarter of the contenders on every hot stamp, and the workgroup's other three wavefronts, idle before, do the work.
  static constexpr int kGroupRows = NVBX_CAM_GR, kGroupCols = NVBX_CAM_GC;
  // end point (camera frame) of the ray through the centre of pixel (prow, pcol) at depth `de` along the optical axis
  __device__ void ray_end(const Frame& f, int prow, int pcol, float de, float* pc) const {
    const float rx = (((float)pcol + 0.5f) - f.cu) / f.fu;
    const float ry = (((float)prow + 0.5f) - f.cv) / f.fv;
    pc[0] = de * rx; pc[1] = de * ry; pc[2] = de;
  }
  // measured depth at the voxel centre `pc` and the voxel's own depth; 1 = update, 0 = voxel not touched,
  // -1 = the voxel projects onto invalid depth (weight decays if invalid_depth_decay_factor >= 0)
  template <typename Img>
  __device__ int sample(const Frame& f, const Img& depth, const float* pc, float* ds, float* vd) const {
    float u, v;
    if (!cam_project(f, pc, &u, &v)) return 0;
    *vd = pc[2];
    if (f.max_dist > 0.0f && *vd > f.max_dist) return 0;
    return interp_depth(depth, f.rows, f.cols, u, v, f.interp_nearest, ds);
  }
};

// Lidar: nvbx_lidar_math.h.  el_tab[k] = {sin, cos} of beam row k's elevation, az_tab[j] = {sin, cos} of column j's azimuth.
struct LidarSensor {
  static constexpr bool kLongRays = true;
  // long rays: the walk is a serial chain per ray and the flushes are chains of dependent HBM round trips, so the lever is the number
  // of wavefronts in flight: FEW rays per wavefront, MANY lanes per ray.  A 200 m ray is ~250 dependent block steps; its lanes share
  // it: lane s replays the (cheap, insert-free) traversal up to its segment -- the same float operations in the same order, so the
  // state is bit-identical -- and then walks only its segment with set inserts.  Measured (tools/lidar_tile_sweep.sh, 1024x64 beams,
  // ray subsampling 2, us per scan): 4x4 rays x 4 segments 155 | 2x4x8 113 | 2x2x16 82 | 1x4x16 74 | 1x2x32 75 | 1x1x64 111.
#ifndef NVBX_LIDAR_TR
#define NVBX_LIDAR_TR 1
#define NVBX_LIDAR_TC 2
#define NVBX_LIDAR_SEG 32
#endif
  static constexpr int kTileRows = NVBX_LIDAR_TR, kTileCols = NVBX_LIDAR_TC;      // (tuning knobs: tools/lidar_tile_sweep.sh)
#ifndef NVBX_LIDAR_FR
#define NVBX_LIDAR_FR 6
#define NVBX_LIDAR_PD 4
#endif
#ifndef NVBX_LIDAR_SPARSE_STRIDED
#define NVBX_LIDAR_SPARSE_STRIDED 1       // (0: eight consecutive records per pass -- 126.5 instead of 116.4 us, EXPERIMENTS.md)
#endif
#ifndef NVBX_LIDAR_SET
#define NVBX_LIDAR_SET 1024
#define NVBX_LIDAR_FLUSH 256
#endif
  static constexpr int kSetSize = NVBX_LIDAR_SET, kFlushRounds = NVBX_LIDAR_FR; // early flush at 256 keys: 6 x 64 >= 256 + one step's additions
  static constexpr int kSegments = NVBX_LIDAR_SEG;
  static constexpr int kProbeDepth = NVBX_LIDAR_PD;
  static constexpr bool kRiders = false;
  static constexpr int kThreads = 64;
  static constexpr int kGroupRows = 1, kGroupCols = 1;   // one tile (bundle of rays) per workgroup
  nvbx_lidar_model l;
  const float2* el_tab; const float2* az_tab;
  float max_diff_m, max_ray_dist_m;
  __device__ void beam_dir(int row, int col, float* d) const {
    const float2 e = el_tab[row], a = az_tab[col];
    d[0] = e.y * a.y; d[1] = e.y * a.x; d[2] = e.x;
  }
  __device__ void ray_end(const Frame&, int prow, int pcol, float de, float* pc) const {
    float d[3]; beam_dir(prow, pcol, d);
    pc[0] = de * d[0]; pc[1] = de * d[1]; pc[2] = de * d[2];
  }
  // [U] interpolateLidarImage restated: bilinear if the four beams are valid and agree within max_diff_m, else the
  // nearest beam if the voxel centre lies within max_ray_dist_m of that beam's ray.  Depth = range along the beam.
  template <typename Img>
  __device__ int sample(const Frame& f, const Img& img, const float* pc, float* ds, float* vd) const { int px; return sample_px(f, img, pc, ds, vd, &px); }
  // the same, also reporting which rule measured: *nearest_px = pixel index (row * cols + col) of the beam the nearest-beam rule used, -1 otherwise
  template <typename Img>
  __device__ int sample_px(const Frame& f, const Img& img, const float* pc, float* ds, float* vd, int* nearest_px) const {
    *nearest_px = -1;
    const float r = nvbx_lidar_range(pc);
    *vd = r;
    if (f.max_dist > 0.0f && r > f.max_dist) return 0;      // (before the projection: it costs two atan2)
    float u, v;
    if (!nvbx_lidar_project(&l, pc, r, &u, &v)) return 0;
    const float uc = u - 0.5f, vc = v - 0.5f;
    const float fx = floorf(uc), fy = floorf(vc);
    const int x0 = (int)fx, y0 = (int)fy;
    if (!(x0 < 0 || y0 < 0 || x0 + 1 > f.cols - 1 || y0 + 1 > f.rows - 1)) {
      const int32_t i00 = pix(y0, x0, f.cols);
      const float f00 = img(i00), f10 = img(i00 + 1), f01 = img(i00 + f.cols), f11 = img(i00 + f.cols + 1);
      __builtin_amdgcn_sched_barrier(0);      // both rows' loads in flight before the first tap is looked at (else: two serial round trips)
      if (f00 > 0.0f && f10 > 0.0f && f01 > 0.0f && f11 > 0.0f) {
        const float mx = fmaxf(fmaxf(f00, f10), fmaxf(f01, f11)), mn = fminf(fminf(f00, f10), fminf(f01, f11));
        if (mx - mn <= max_diff_m) {
          const float ax = uc - fx, ay = vc - fy;
          const float top = __builtin_fmaf(ax, f10, (1.0f - ax) * f00);
          const float bot = __builtin_fmaf(ax, f11, (1.0f - ax) * f01);
          *ds = __builtin_fmaf(ay, bot, (1.0f - ay) * top);
          return 1;
        }
      }
    }
    const int c = (int)floorf(u), rr = (int)floorf(v);
    if (c < 0 || rr < 0 || c >= f.cols || rr >= f.rows) return 0;
    const float d = img(pix(rr, c, f.cols));
    if (!(d > 0.0f)) return 0;
    float dir[3]; beam_dir(rr, c, dir);
    const float dot = __builtin_fmaf(pc[2], dir[2], __builtin_fmaf(pc[1], dir[1], pc[0] * dir[0]));
    const float ex = __builtin_fmaf(-dot, dir[0], pc[0]), ey = __builtin_fmaf(-dot, dir[1], pc[1]), ez = __builtin_fmaf(-dot, dir[2], pc[2]);
    // (squared distances compared: one IEEE square root less per voxel on the VALU-bound LiDAR path; the oracle does the same)
    if (__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) > max_ray_dist_m * max_ray_dist_m) return 0;
    *ds = d;
    *nearest_px = pix(rr, c, f.cols);
    return 1;
  }
};

constexpr int LSET_FLUSH = NVBX_LIDAR_FLUSH;     // early-flush threshold (long rays): keeps the 1024-entry set <= ~30 % full, probes short


// Claim an entry's stamp word for (frame, camera bit); `cur` = the word as last seen.  True iff THIS call moved the entry to the
// frame (the caller then appends the block to the view list exactly once); otherwise it only makes sure the camera's bit is set.
__device__ inline bool stamp_claim(uint32_t* p, uint32_t cur, uint32_t frame_id, uint32_t cam_bit) {
  const uint32_t want = (frame_id << 8) | cam_bit;
  for (;;) {
    if (stamp_frame(cur) == frame_id) { if (!(cur & cam_bit)) atomicOr(p, cam_bit); return false; }
    const uint32_t old = atomicCAS(p, cur, want);
    if (old == cur) return true;
    cur = old;                          // another tile got there first (or `cur` was a guess): look again
  }
}
// One block key -> HBM: insert-if-absent, stamp the entry with this frame (and this camera's bit), and report whether THIS call was
// the first of the frame to do so (the caller then appends {slot, x, y, z} to the view list exactly once).  The common case -- the
// block exists and a neighbouring tile has stamped it already -- is ONE 16-B load: key, slot and stamp arrive together.
// `entry_out` (optional): the block's hash entry whenever it exists after the call (-1: table full) -- also when this call was not the
// first, so that a caller that needs the slot of a block ANOTHER thread of the same launch has just inserted never has to look the key
// up again with plain loads (hash_find may read a stale EMPTY from L1 / a non-coherent L2 and report "absent").
__device__ inline bool mark_block(const DMap& m, u64 key, uint32_t frame_id, uint32_t cam_bit, int4* rec_out, int32_t* entry_out = nullptr) {
  int32_t x, y, z; unpack_key(key, &x, &y, &z);
  uint32_t h = table_pos(m, x, y, z);
  uint32_t slot = SLOT_INVALID, cur = STAMP_NEVER;
  bool found = false;
  if (entry_out) *entry_out = -1;
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&m.table[h]);
    const u64 k = ((u64)e.y << 32) | (u64)e.x;
    if (k == key) { if (entry_out) *entry_out = (int32_t)h; if (stamp_frame(e.w) == frame_id && (e.w & cam_bit)) return false; slot = e.z; cur = e.w; found = true; break; }
    if (k == KEY_EMPTY) break;           // (may be a stale EMPTY: hash_insert's CAS is the truth)
    h = (h + 1) & m.mask;
  }
  if (!found) {
    bool is_new;
    const int32_t hi = hash_insert(m, x, y, z, F_TSDF, &is_new);
    if (hi < 0) return false;
    h = (uint32_t)hi;
    if (entry_out) *entry_out = hi;
  }
  if (!stamp_claim(&m.table[h].stamp, cur, frame_id, cam_bit)) return false;
  while (slot == SLOT_INVALID) slot = ld_slot_acquire(&m.table[h]);     // the inserting lane publishes right after its CAS
  *rec_out = make_int4((int32_t)slot, x, y, z);
  return true;
}

// wave-aggregated append of this lane's record (if `first`) to the frame's view list: one returning atomic per wave
__device__ inline void view_append(int32_t* cnt, int4* view_list, int32_t list_cap, bool first, int4 rec, int lane) {
  const u64 mask = __ballot(first);
  if (!mask) return;
  int32_t base = 0;
  const int leader = __ffsll((long long)mask) - 1;
  if (lane == leader) base = atomicAdd(cnt, (int32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (first) {
    const int32_t pos = base + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (pos < list_cap) view_list[pos] = rec;
  }
}

// [U] workspace bounds of the view calculator (workspace_bounds_type, mapper_initialization.cpp:337-358): a block is kept
// iff its cube overlaps the bounds (height bounds: z only)
__device__ inline bool block_in_workspace(const Frame& f, int32_t bx, int32_t by, int32_t bz) {
  if (f.ws_type == 0) return true;
  const int32_t cur[3] = {bx, by, bz};
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (f.ws_type == 1 && a < 2) continue;
    const float lo = (float)cur[a] * f.block_size, hi = (float)(cur[a] + 1) * f.block_size;
    if (!(hi > f.ws_min[a]) || !(lo < f.ws_max[a])) ok = false;
  }
  return ok;
}
// insert `key` into the tile's LDS set; false = probe window exhausted (caller sends the key to HBM itself)
template <int LSET>
__device__ inline bool lset_insert(u64* lset, int32_t bx, int32_t by, int32_t bz, u64 key, bool* added) {
  static_assert((LSET & (LSET - 1)) == 0, "power of two");
  const uint32_t lh = ((index_hash(bx, by, bz) * 2654435761u) >> 16) & (LSET - 1);
  *added = false;
#pragma unroll 1
  for (int p = 0; p < 16; p++) {
    const u64 old = atomicCAS(&lset[(lh + p) & (LSET - 1)], KEY_EMPTY, key);
    if (old == KEY_EMPTY) { *added = true; return true; }
    if (old == key) return true;
  }
  return false;
}
// Amanatides-Woo: advance to the next block along the ray (select without dynamic register indexing)
// Amanatides-Woo through the block grid with the crossing parameters in CLOSED FORM: crossing number k of axis a lies at
//   T_a(k) = fmaf(k, tdelta_a, tmax0_a)            (one rounding; the checker evaluates the same fmaf: oracle/nvblox_oracle.c raycast_blocks)
// instead of tmax_a accumulated by k additions.  Same traversal up to the last bit of a near-tie -- and a state that depends on the crossing
// COUNTS (n_x, n_y, n_z) alone, so a lane can enter the traversal at any step in O(1) (dda_jump) instead of replaying every step before it:
// a LiDAR lane used to replay up to 234 steps of a 200 m ray before its own 16 (35 of the slowest bundle's 78 us, tools/wg_timeline_lidar.py).
// A step: the axis with the smallest next crossing (ties: x before y before z), select-only.
struct Dda { int32_t cur[3], step[3], n[3]; float t0[3], dt[3], tm[3]; };
__device__ inline float dda_T(const Dda& d, int a, int32_t k) { return __builtin_fmaf((float)k, d.dt[a], d.t0[a]); }
__device__ inline void dda_step(Dda& d) {
  const bool s1 = d.tm[1] < d.tm[0];
  const float m01 = s1 ? d.tm[1] : d.tm[0];
  const bool s2 = d.tm[2] < m01;
  const bool a0 = !s1 && !s2, a1 = s1 && !s2;
  d.n[0] += a0 ? 1 : 0; d.n[1] += a1 ? 1 : 0; d.n[2] += s2 ? 1 : 0;
  d.cur[0] += a0 ? d.step[0] : 0; d.cur[1] += a1 ? d.step[1] : 0; d.cur[2] += s2 ? d.step[2] : 0;
  d.tm[0] = dda_T(d, 0, d.n[0]); d.tm[1] = dda_T(d, 1, d.n[1]); d.tm[2] = dda_T(d, 2, d.n[2]);
}
// undo the last step taken: of the crossings taken, the one with the LARGEST parameter (ties: z before y before x -- the reverse of dda_step's order)
__device__ inline void dda_unstep(Dda& d) {
  const float l0 = d.n[0] > 0 ? dda_T(d, 0, d.n[0] - 1) : -1.0f, l1 = d.n[1] > 0 ? dda_T(d, 1, d.n[1] - 1) : -1.0f, l2 = d.n[2] > 0 ? dda_T(d, 2, d.n[2] - 1) : -1.0f;
  const bool s2 = d.n[2] > 0 && l2 >= l1 && l2 >= l0;
  const bool a1 = !s2 && d.n[1] > 0 && l1 >= l0;
  const bool a0 = !s2 && !a1 && d.n[0] > 0;
  d.n[0] -= a0 ? 1 : 0; d.n[1] -= a1 ? 1 : 0; d.n[2] -= s2 ? 1 : 0;
  d.cur[0] -= a0 ? d.step[0] : 0; d.cur[1] -= a1 ? d.step[1] : 0; d.cur[2] -= s2 ? d.step[2] : 0;
  d.tm[0] = dda_T(d, 0, d.n[0]); d.tm[1] = dda_T(d, 1, d.n[1]); d.tm[2] = dda_T(d, 2, d.n[2]);
}
// Enter the traversal after exactly K steps (from the initial state).  (1) a parameter tau at which about K crossings have happened (the crossing
// density is linear in the parameter); (2) the EXACT state "every crossing with T < tau taken" -- counted per axis with the same fmaf the
// traversal compares, so it is a state the step-by-step traversal passes through whatever the estimate was; (3) a few steps forwards or
// backwards until the count is K.  `inv[a]` = 1 / tdelta_a (0 for an axis the ray does not move along).
__device__ inline void dda_jump(Dda& d, int32_t K, const float* inv) {
  const float s1 = (inv[0] + inv[1]) + inv[2];
  float s0 = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; a++) s0 = s0 + (inv[a] > 0.0f ? 1.0f - d.t0[a] * inv[a] : 0.0f);
  const float tau = s1 > 0.0f ? ((float)K - s0) / s1 : 0.0f;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int32_t k = 0;
    if (inv[a] > 0.0f) {
      const float e = ceilf((tau - d.t0[a]) * inv[a]);
      k = e > 0.0f ? (e < 1.0e6f ? (int32_t)e : 1000000) : 0;
      while (k > 0 && dda_T(d, a, k - 1) >= tau) k--;
      while (k < 1000000 && dda_T(d, a, k) < tau) k++;
    }
    d.n[a] = k; d.cur[a] += k * d.step[a]; d.tm[a] = dda_T(d, a, k);
  }
  int32_t have = (d.n[0] + d.n[1]) + d.n[2];
  while (have < K) { dda_step(d); have++; }
  while (have > K) { dda_unstep(d); have--; }
}
// The ray of depth pixel (prow, pcol) with measured depth `d` through the block grid: traversal state at the sensor's block, number of block
// steps to the block of the end point min(d + truncation, max integration distance) (-1: no ray -- inactive lane or invalid depth), 1 / tdelta.
template <typename Sensor>
__device__ inline int32_t view_ray_setup(const Frame& f, const Sensor& sensor, bool& active, float d, int prow, int pcol, Dda& dd, float* inv_dt) {
  int32_t nsteps = -1;
  if (active) {
    if (!(d > 0.0f)) active = false;
    else {
      float de = d + f.trunc;
      if (f.max_dist > 0.0f && de > f.max_dist) de = f.max_dist;
      float pc[3], pl[3];
      sensor.ray_end(f, prow, pcol, de, pc);
      apply_rt(f.R_LC, f.t_LC, pc[0], pc[1], pc[2], pl);
      nsteps = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float s = f.t_LC[a] / f.block_size, t = pl[a] / f.block_size;
        dd.cur[a] = (int32_t)floorf(s);
        const int32_t end = (int32_t)floorf(t);
        const int32_t db = end - dd.cur[a]; nsteps += db < 0 ? -db : db;
        const float ray = t - s;
        dd.step[a] = ray > 0.0f ? 1 : (ray < 0.0f ? -1 : 0);
        const float corrected = dd.step[a] > 0 ? 1.0f : 0.0f;
        const float dist_to_boundary = corrected - (s - (float)dd.cur[a]);
        if (fabsf(ray) < 1e-9f) { dd.t0[a] = 2.0f; dd.dt[a] = 2.0f; }
        else { dd.t0[a] = dist_to_boundary / ray; dd.dt[a] = (float)dd.step[a] / ray; inv_dt[a] = fabsf(ray); }
        dd.tm[a] = dd.t0[a];
      }
    }
  }
  return nsteps;
}
// Flush: compact the set (ballot + popcount), then every key goes to HBM with the dependent round trips taken
// PHASE-WISE over up to R keys per lane at once: (A) the first PD probe positions of every key are loaded together
// (2 cover ~98 % of lookups at a room-sized map's load factor, 4 are used for the larger LiDAR maps), (B) resolved -- a key further down its probe chain, a new block, or a
// slot not published yet takes the general mark_block path, (C) the frame-stamp exchanges of all keys not yet stamped
// are issued together, (D) ONE wave-aggregated returning atomicAdd reserves view-list space for all first-stampers,
// (E) records are stored.  A camera tile flushes ~60 keys in one such pass; a long LiDAR bundle 256+ keys per pass
// instead of 64 per dependent round.  Whole wave must call.
// NW = wavefronts of the workgroup that share the set (camera: 4 tiles per workgroup; LiDAR: 1): wave w compacts the w-th part of the
// set, the parts' counts meet in LDS (s_part), and key number i of the compacted list goes to thread i of the workgroup.
template <int LSET, int R, int PD, int NW = 1>
__device__ inline void flush_set(const DMap& m, const Frame& f, u64* lset, u64* lkeys, int32_t* cnt, int4* view_list, int32_t list_cap,
                                 int lane, bool clear, int32_t* s_part = nullptr) {
  __syncthreads();
  const int wave = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
  constexpr int PART = LSET / NW;
  static_assert(PART % 64 == 0, "whole wavefronts per part");
  int32_t nk = 0;
  if (NW > 1) {              // counts first: where this wave's keys go depends on the parts before it
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < PART / 64; i++) c += (int32_t)__popcll(__ballot(lset[wave * PART + i * 64 + lane] != KEY_EMPTY));
    if (lane == 0) s_part[wave] = c;
    __syncthreads();
    int32_t before = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { const int32_t cw = s_part[w]; if (w < wave) before += cw; nk += cw; }
    int32_t pos = before;
#pragma unroll
    for (int i = 0; i < PART / 64; i++) {
      const u64 kk = lset[wave * PART + i * 64 + lane];
      const u64 mask = __ballot(kk != KEY_EMPTY);
      if (kk != KEY_EMPTY) lkeys[pos + (int32_t)__popcll(mask & ((1ull << lane) - 1ull))] = kk;
      pos += (int32_t)__popcll(mask);
      if (clear) lset[wave * PART + i * 64 + lane] = KEY_EMPTY;
    }
  } else {
#pragma unroll
    for (int i = 0; i < LSET / 64; i++) {
      const u64 kk = lset[i * 64 + lane];
      const u64 mask = __ballot(kk != KEY_EMPTY);
      if (kk != KEY_EMPTY) lkeys[nk + (int32_t)__popcll(mask & ((1ull << lane) - 1ull))] = kk;
      nk += (int32_t)__popcll(mask);
      if (clear) lset[i * 64 + lane] = KEY_EMPTY;
    }
  }
  __syncthreads();
  NVBX_T(0, 3);
#ifndef NVBX_WGT_WALK_START
  if (NW > 1) NVBX_TV(0, 6, nk);
#endif
  // key number kb + r * (NW * 64) + (this thread's number in the workgroup): wave w takes the w-th 64 keys of every round
  const int tlane = NW > 1 ? (int)threadIdx.x : lane;
  for (int32_t kb = 0; kb < nk; kb += R * NW * 64) {         // one pass per R x NW x 64 keys (workgroup-uniform)
    const int rounds = min(R, (nk - kb + NW * 64 - 1) / (NW * 64));
    u64 key[R]; uint32_t h[R]; uint4 e[R][PD]; bool have[R];
    // (A) the first PD probe positions of every key, all in flight
#pragma unroll
    for (int r = 0; r < R; r++) {
      have[r] = r < rounds && (kb + r * NW * 64 + tlane) < nk;
      key[r] = have[r] ? lkeys[kb + r * NW * 64 + tlane] : KEY_EMPTY;
      int32_t x, y, z; unpack_key(key[r], &x, &y, &z);
      h[r] = have[r] ? table_pos(m, x, y, z) : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; r++) if (r < rounds) {
#pragma unroll
      for (int q = 0; q < PD; q++) e[r][q] = *reinterpret_cast<const uint4*>(&m.table[(h[r] + q) & m.mask]);
    }
    // (B) resolve + (C) stamp exchanges in flight
    bool fast[R], first[R], claim[R], ins[R]; uint32_t old[R], seen[R], slot[R], hpos[R]; int4 rec[R];
    const uint32_t want = (f.frame_id << 8) | f.cam_bit;
    bool any_ins = false;
#pragma unroll
    for (int r = 0; r < R; r++) {
      fast[r] = false; first[r] = false; claim[r] = false; ins[r] = false; old[r] = 0u; seen[r] = 0u; hpos[r] = 0u; slot[r] = SLOT_INVALID; rec[r] = make_int4(0, 0, 0, 0);
      if (r < rounds && have[r]) {
        uint32_t hh = h[r], st = 0; bool open = true, hit = false;        // open: no EMPTY entry seen yet on the probe chain
        int qe = -1;                                                      // first EMPTY position of the chain, if the key is not in front of it
#pragma unroll
        for (int q = 0; q < PD; q++) {
          const u64 kq = ((u64)e[r][q].y << 32) | (u64)e[r][q].x;
          if (open && !fast[r] && kq == key[r]) { slot[r] = e[r][q].z; st = e[r][q].w; hh = (h[r] + q) & m.mask; fast[r] = true; hit = true; }
          if (open && kq == KEY_EMPTY) { open = false; if (!hit) qe = q; }
        }
        if (slot[r] == SLOT_INVALID) fast[r] = false;              // being inserted right now: general path waits for the slot
        if (fast[r]) {
          hpos[r] = hh; seen[r] = st;
          if (stamp_frame(st) != f.frame_id) { claim[r] = true; old[r] = atomicCAS(&m.table[hh].stamp, st, want); }   // the returning atomics of a pass: in flight together
          else if (!(st & f.cam_bit)) atomicOr(&m.table[hh].stamp, f.cam_bit);     // stamped by another camera of this batch: add our bit (not waited for)
#ifndef NVBX_NO_BATCH_INSERT             // (A/B: tools/build_variant.sh nobatch "-DNVBX_NO_BATCH_INSERT")
        } else if (qe >= 0) { ins[r] = true; any_ins = true; hpos[r] = (h[r] + qe) & m.mask;        // a NEW block (as far as this pass can see)
#endif
        }
      }
    }
    // (B') new blocks, batch-wise: the pass's key inserts in flight together, then ONE pop of the free stack for all the wavefront's winners
    // (hash_insert pops one slot per block: two returning atomics on ONE address each -- free-stack top and high-water mark --, ~12 ns apiece
    // chip-wide, i.e. 2.7 ms of a first LiDAR scan's 112 k new blocks before anything else; and a chain of ~8 dependent round trips per key, R
    // keys one after the other).  A lane that loses its insert (another wavefront's key landed in the entry first) takes the general path.
    bool won[R];
#pragma unroll
    for (int r = 0; r < R; r++) won[r] = false;
    if (__ballot(any_ins)) {
      u64 oldk[R];
#pragma unroll
      for (int r = 0; r < R; r++) if (ins[r]) oldk[r] = atomicCAS(&m.table[hpos[r]].key, KEY_EMPTY, key[r]);
      int32_t wtotal = 0, wpre[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        won[r] = ins[r] && oldk[r] == KEY_EMPTY;
        const u64 mask = __ballot(won[r]);
        wpre[r] = wtotal + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
        wtotal += (int32_t)__popcll(mask);
      }
      if (wtotal) {
        int32_t top = 0;
        if (lane == 0) {
          top = atomicSub(&m.counters[C_FREE_TOP], wtotal);
          if (top < wtotal) { atomicAdd(&m.counters[C_FREE_TOP], wtotal - (top > 0 ? top : 0)); atomicExch(&m.counters[C_OVERFLOW], 1); }     // pool exhausted: give back what was not there
        }
        top = __shfl(top, 0);
        uint32_t ost[R];
#pragma unroll
        for (int r = 0; r < R; r++) if (won[r]) {
          const int32_t idx = top - 1 - wpre[r];
          slot[r] = idx >= 0 ? m.free_stack[idx] : SLOT_NONE;
          ost[r] = atomicCAS(&m.table[hpos[r]].stamp, STAMP_NEVER, want);          // (a fresh entry's stamp; somebody may have met the key and claimed it already)
        }
        int32_t hwm = 0;
#pragma unroll
        for (int r = 0; r < R; r++) if (won[r]) {
          int32_t x, y, z; unpack_key(key[r], &x, &y, &z);
          if (slot_ok(slot[r])) {
            m.slot_index[3 * slot[r]] = x; m.slot_index[3 * slot[r] + 1] = y; m.slot_index[3 * slot[r] + 2] = z;
            m.slot_entry[slot[r]] = hpos[r];
            atomicOr(&m.slot_flags[slot[r]], F_TSDF);
            hwm = max(hwm, (int32_t)slot[r] + 1);
          }
          __hip_atomic_store(&m.table[hpos[r]].slot, slot[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // published: every winner of the pass BEFORE this wavefront waits for anybody else's
          first[r] = ost[r] == STAMP_NEVER || stamp_claim(&m.table[hpos[r]].stamp, ost[r], f.frame_id, f.cam_bit);
          rec[r] = make_int4((int32_t)slot[r], x, y, z);
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) hwm = max(hwm, __shfl_xor(hwm, o));
        if (lane == 0 && hwm) atomicMax(&m.counters[C_HIGH_WATER], hwm);
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (r < rounds && have[r]) {
        if (fast[r]) {
          first[r] = claim[r] && (old[r] == seen[r] || stamp_claim(&m.table[hpos[r]].stamp, old[r], f.frame_id, f.cam_bit));   // (a lost CAS: another tile claimed it, add our bit)
          if (first[r]) { int32_t x, y, z; unpack_key(key[r], &x, &y, &z); rec[r] = make_int4((int32_t)slot[r], x, y, z); }
        } else if (!won[r]) {
          first[r] = mark_block(m, key[r], f.frame_id, f.cam_bit, &rec[r]);     // longer probe chain, a lost insert, or slot not published yet
        }
      }
    }
    NVBX_T(0, 4);
    // (D) one reservation for the whole pass
    int32_t total = 0; int32_t pre[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const u64 mask = (r < rounds) ? __ballot(first[r]) : 0ull;
      pre[r] = total + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
      total += (int32_t)__popcll(mask);
    }
    if (total) {
      int32_t base = 0;
      if (lane == 0) base = atomicAdd(cnt, total);
      base = __shfl(base, 0);
      // (E)
#pragma unroll
      for (int r = 0; r < R; r++) if (r < rounds && first[r]) { const int32_t pos = base + pre[r]; if (pos < list_cap) view_list[pos] = rec[r]; }
      NVBX_T(0, 5);
    }
  }
  __syncthreads();
}

// Workgroups [0, n_edt_wg) (camera launches only, when an EDT was held back by updateEsdf) are EDT workers with all four
// wavefronts -- dispatched first: the EDT is the longer chain; the workgroups after them mark the view (first wavefront only).
// The frames of one launch set: ONE depth frame, or a batch of up to MAX_BATCH camera frames of the same image size that
// nvbx_integrate_depth_batch integrates with one view-marking launch and one TSDF-update launch (the reference feeds up to four
// cameras through one mapper, one integrateDepth call each: nvblox_node.hpp:298-332).  Kernel argument (SGPRs / scalar loads).
template <typename Img, int NB> struct FrameSet { Frame f[NB]; Img img[NB]; int32_t n; };

// workgroups of one frame's view marking: its tile groups, padded to a multiple of the XCD count (XCD-banded numbering in the kernel)
template <typename Sensor> static int mark_view_tile_wgs(const Frame& f) {
  const int tiles_x = (f.n_ray_cols + Sensor::kTileCols - 1) / Sensor::kTileCols, tiles_y = (f.n_ray_rows + Sensor::kTileRows - 1) / Sensor::kTileRows;
  const int n_groups = ((tiles_x + Sensor::kGroupCols - 1) / Sensor::kGroupCols) * ((tiles_y + Sensor::kGroupRows - 1) / Sensor::kGroupRows);
  return NSH * ((n_groups + NSH - 1) / NSH);
}
template <typename Sensor> static size_t mark_view_smem(bool edt_rides) {
  const size_t mark = 2 * (size_t)Sensor::kSetSize * sizeof(u64);
  return (Sensor::kRiders && edt_rides && sizeof(EdtShared) > mark) ? sizeof(EdtShared) : mark;
}
// Occupancy of the two fused launches, by batch size (the attribute's arguments depend on the template parameter).  One camera frame launches ~960
// workgroups -- fewer than are resident at the compiler's own register choice (87 VGPRs = 5 waves per SIMD = 1 280 workgroups of four wavefronts), and
// squeezing it costs time (8 waves per SIMD asked for: 11.2 -> 13.1 us).  A batch of eight launches 2 256: the tiles, dispatched behind the riders,
// started when the first 1 280 workgroups were done (11-15 us into a 31 us launch, tools/wg_timeline_batch.py) -- there 8 waves per SIMD (64 VGPRs,
// 26 spilled to scratch) are worth it: 31.6 -> 26.9 us; the fused TSDF / colour launch likewise (three 8-wavefront workgroups per CU -> four): 30.5 -> 28.6 us.
#ifndef NVBX_MARK_VIEW_ATTR
#define NVBX_MARK_VIEW_ATTR __attribute__((amdgpu_waves_per_eu(NB > 1 ? 8 : 1, NB > 1 ? 8 : 8)))
#endif
#ifndef NVBX_FUSED_ATTR
#define NVBX_FUSED_ATTR __attribute__((amdgpu_waves_per_eu(NB > 1 ? 8 : 1, NB > 1 ? 8 : 8)))
#endif
// The launch's body as a function of the workgroup's NUMBER (`wg_index`, not blockIdx.x): k_mark_view passes blockIdx.x; k_mark_view_pair (round 6) runs the
// bodies of TWO mappers' view-marking launches in one grid -- the second mapper's workgroups are numbered from its own 0 (every part's count is a multiple
// of 8, so blockIdx.x & 7 -- the shard of the sharded counters, my_shard() -- is also wg_index & 7).
template <typename Img, typename Sensor, int NB>
__device__ __forceinline__ void mark_view_body(const DMap& m, const FrameSet<Img, NB>& fs, const Sensor& sensor, int4* view_list, int32_t list_cap,
                                               int32_t reset_esdf_dirty, int32_t n_edt_wg, const EsdfArgs& ea, const TraceRiderT<NB>& tr, const int32_t wg_index, unsigned char* smem) {
  constexpr int LSET = Sensor::kSetSize, FR = Sensor::kFlushRounds;
  int32_t tile_wg = wg_index;      // this workgroup's number among the tiles
  NVBX_T(0, 0);
  // this launch has STARTED, so every launch enqueued before it on the stream has finished -- among them the tr.fence_report colour-reading launches
  // whose images' frames wait for exactly this news (frames.hip)
  if (wg_index == 0 && threadIdx.x == 0) __hip_atomic_store(&m.host_mirror[4], tr.fence_report, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (Sensor::kRiders) {
    // riders: [EDT workers][sphere-tracing workers of a held-back colour frame (colour deferral)] -- before the tiles, or
    // (tr.n_tile_wg > 0) after them.  All counts are multiples of 8, so a workgroup's XCD (blockIdx.x & 7) is also its number's & 7.
    const int32_t rider = tr.n_tile_wg > 0 ? wg_index - tr.n_tile_wg : wg_index;
    const bool is_rider = tr.n_tile_wg > 0 ? rider >= 0 : rider < n_edt_wg + tr.n_wg + tr.n_scan_wg + tr.n_mark_wg;
    if (is_rider) {
      if (Sensor::kThreads > 256 && threadIdx.x >= 256) return;      // (a rider is a 256-thread worker: the wavefronts a wider tile group needs go home at once -- they are not waited for by the others' barriers)
      if (rider < n_edt_wg) esdf_edt_worker(m, ea, (int)rider, n_edt_wg, reinterpret_cast<EdtShared*>(smem));
      // (sphere tracing: all four wavefronts; independent of the view marking -- it reads the TSDF and the insert-only hash, and new entries point at all-zero blocks)
      else if (rider < n_edt_wg + tr.n_wg) {
        const int tw = (int)(rider - n_edt_wg);
        if (NB > 1 && tr.lanes == 2) sphere_trace_worker<NB, 2>(m, tr.ps, tr.synth, tr.srows, tr.scols, tr.max_steps, tr.max_len, tr.eps_m, tw);
        else if (tr.lanes == 4) sphere_trace_worker<NB, 4>(m, tr.ps, tr.synth, tr.srows, tr.scols, tr.max_steps, tr.max_len, tr.eps_m, tw);
        else sphere_trace_worker<NB, 8>(m, tr.ps, tr.synth, tr.srows, tr.scols, tr.max_steps, tr.max_len, tr.eps_m, tw);
      }
      // (candidate discovery of the held-back colour frame(s), for the fused colour + TSDF launch that follows: four wavefronts of 64 slots each)
      else if (rider < n_edt_wg + tr.n_wg + tr.n_scan_wg)
        color_scan_worker<NB>(m, tr.ps, tr.cand, tr.cand_cnt_idx, tr.cand_reset_idx, (int)(rider - n_edt_wg - tr.n_wg) * 4 + (int)(threadIdx.x >> 6), tr.n_scan_wg * 4);
      // (ESDF site marking of the held-back update, first wavefront only: it reads the TSDF as the last update left it -- nothing in this launch
      //  writes voxels -- and allocates ESDF blocks beside the view marking's TSDF blocks; `ea` is its argument then: no EDT rides, n_edt_wg = 0)
      // (all four wavefronts are workers -- a frame dirties ~300 blocks, 4 x 256 workers take at most one entry each: an entry is a chain of
      //  dependent round trips, and a worker with two of them was the launch's tail; the workgroup then counts itself in as one arrival)
      else {
        const int w = (int)(rider - n_edt_wg - tr.n_wg - tr.n_scan_wg);
        esdf_mark_worker(m, ea, w * 4 + (int)(threadIdx.x >> 6), tr.n_mark_wg * 4);
        if (ea.self_reset) { __syncthreads(); if (threadIdx.x < 64) esdf_mark_pass_done(m, ea, tr.n_mark_wg, w); }
      }
      if (rider >= n_edt_wg) NVBX_INV_TSDF_READER(m);       // (sphere tracing, candidates, marking: still no TSDF writer beside them when they end)
      NVBX_T(0, 7);
      return;
    }
    if (tr.n_tile_wg == 0) tile_wg -= n_edt_wg + tr.n_wg + tr.n_scan_wg + tr.n_mark_wg;
  }
  __shared__ int32_t s_part[8];
  u64* lset = reinterpret_cast<u64*>(smem);
  u64* lkeys = lset + LSET;
  const int lane = threadIdx.x & 63;
  constexpr int TR = Sensor::kTileRows, TC = Sensor::kTileCols, NSEG = Sensor::kSegments;
  constexpr int GR = Sensor::kGroupRows, GC = Sensor::kGroupCols, NW = GR * GC;       // tiles (wavefronts) per workgroup
  static_assert(TR * TC * NSEG <= 64, "one wavefront per tile");
  static_assert(NW * 64 == Sensor::kThreads && NW <= 8, "one wavefront per tile of the group");
  const int wave = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
  const Frame& f0 = fs.f[0];                    // (image size, subsampling and view frame id are the same for every frame of a batch)
  const int tiles_x = (f0.n_ray_cols + TC - 1) / TC, tiles_y = (f0.n_ray_rows + TR - 1) / TR;
  const int groups_x = (tiles_x + GC - 1) / GC, groups_y = (tiles_y + GR - 1) / GR;
  // XCD-aware numbering: workgroups go round-robin over the 8 XCDs (each with its own L2), so the tile groups of one XCD (wg & 7) are a
  // contiguous band of group rows -- neighbouring tiles share most of their blocks, i.e. their hash lines (n_edt_wg is a multiple of 8)
  const int wg_all = (int)tile_wg;
  const int n_groups = groups_x * groups_y, per_xcd = (n_groups + NSH - 1) / NSH;
  const int cam = NB > 1 ? wg_all / (NSH * per_xcd) : 0;        // batch: NSH * per_xcd workgroups per camera, camera after camera
  const int wg = wg_all - cam * (NSH * per_xcd);
  const Frame& f = fs.f[cam < fs.n ? cam : 0];
  const Img& depth = fs.img[cam < fs.n ? cam : 0];
  const int group = (wg & (NSH - 1)) * per_xcd + (wg >> 3);
  const int gy = group / groups_x, gx = group - gy * groups_x;
  const int ty = gy * GR + wave / GC, tx = gx * GC + wave % GC;
  const bool tile_ok = (wg >> 3) < per_xcd && group < n_groups && cam < fs.n && ty < tiles_y && tx < tiles_x;
  const int ray = lane / NSEG, seg = lane % NSEG;
  const int ri = ty * TR + ray / TC, ci = tx * TC + ray % TC;
  bool active = tile_ok && ray < TR * TC && ri < f.n_ray_rows && ci < f.n_ray_cols;
  // the ray's depth pixel is requested first: its HBM round trip overlaps the LDS set initialisation
  int prow = ri * f.subsample; if (prow >= f.rows) prow = f.rows - 1;
  int pcol = ci * f.subsample; if (pcol >= f.cols) pcol = f.cols - 1;
  const float d = active ? depth(pix(prow, pcol, f.cols)) : 0.0f;
  for (int i = (int)threadIdx.x; i < LSET; i += NW * 64) lset[i] = KEY_EMPTY;
  if (wg_all == 0 && threadIdx.x == 0) m.counters[C_VIEW_COUNT + ((f.frame_id + 1) & 3)] = 0;   // next frame's counter
  if (Sensor::kLongRays && wg_all == 0 && threadIdx.x < NSH) { *shc_at(m, S_LIDAR_SPARSE, threadIdx.x, 0) = 0; *shc_at(m, S_LIDAR_SPARSE, threadIdx.x, 1) = 0; }     // (field 1: the dense launch's work list, filled by the beam-centric one)
  // an ESDF dirty list already consumed by a marking pass (fused into integrateColor) is emptied before k_integrate_tsdf appends
  if (reset_esdf_dirty && wg_all == 0 && threadIdx.x < NSH) *shc_at(m, S_LIST_ESDF_DIRTY, threadIdx.x, 0) = 0;
  __syncthreads();
  NVBX_T(0, 1);

  Dda dd{};                                     // traversal state of this lane's ray (cur = block, n = crossings taken per axis)
  float inv_dt[3] = {0.0f, 0.0f, 0.0f};          // 1 / tdelta per axis (= |ray| in blocks), for dda_jump
  const int32_t nsteps = view_ray_setup(f, sensor, active, d, prow, pcol, dd, inv_dt);
  // this lane's share of the ray: steps [k0, k1]; the traversal is ENTERED at step k0 (dda_jump: no replay of the steps before it)
  int32_t k0 = 0, k1 = nsteps;
  if (NSEG > 1 && nsteps >= 0) {
    const int32_t q = (nsteps + NSEG) / NSEG;                  // ceil((nsteps + 1) / NSEG)
    k0 = seg * q; k1 = min(nsteps, k0 + q - 1);
    if (k0 > nsteps) k1 = -1;                                   // short ray: nothing left for this segment
    else if (k0 > 0) dda_jump(dd, k0, inv_dt);
  }
  int32_t* cnt = &m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
#ifdef NVBX_WGT_WALK_START
  NVBX_TV(0, 6, wall_clock64() + (unsigned long long)(nsteps & 0));        // (experiment: when the ray set-up is done -- the depth pixel has arrived)
#endif
  if (!Sensor::kLongRays) {
    // camera: a tile's rays cross < 100 blocks in ~20 steps -- walk every ray to its end, then flush once
    for (int32_t k = k0; k <= k1; k++) {                   // (this lane's segment of the ray; the whole ray if it is not shared)
      const u64 key = pack_key(dd.cur[0], dd.cur[1], dd.cur[2]);
      const bool inside = block_in_workspace(f, dd.cur[0], dd.cur[1], dd.cur[2]);
      const uint32_t lh = ((index_hash(dd.cur[0], dd.cur[1], dd.cur[2]) * 2654435761u) >> 16) & (LSET - 1);
      // first probe issued, the traversal step runs in the shadow of the LDS round trip, then the result is looked at
      u64 old = KEY_EMPTY;
      if (inside) old = atomicCAS(&lset[lh], KEY_EMPTY, key);
      dda_step(dd);
      bool spill = false;
      if (inside && old != KEY_EMPTY && old != key) {          // occupied by another block: continue along the probe window
        spill = true;
#pragma unroll 1
        for (int p = 1; p < 16; p++) {
          const u64 o2 = atomicCAS(&lset[(lh + p) & (LSET - 1)], KEY_EMPTY, key);
          if (o2 == KEY_EMPTY || o2 == key) { spill = false; break; }
        }
      }
      if (__ballot(spill)) {                     // probe window exhausted (rare): this key goes to HBM directly
        int4 rec = make_int4(0, 0, 0, 0);
        const bool first = spill && mark_block(m, key, f.frame_id, f.cam_bit, &rec);
        view_append(cnt, view_list, list_cap, first, rec, lane);
      }
    }
    NVBX_T(0, 2);
    flush_set<LSET, FR, Sensor::kProbeDepth, NW>(m, f, lset, lkeys, cnt, view_list, list_cap, lane, false, s_part);
    NVBX_T(0, 7);
    return;
  }
  // LiDAR: hundreds of steps per ray and little sharing at long range -- wave-uniform loop, flush whenever the set is
  // half full
  int32_t nset = 0;                                   // keys in the LDS set (wave-uniform)
#ifdef NVBX_WG_TIMES
  unsigned long long t_flush = 0, n_flush = 0, n_keys = 0;     // (tools/wg_timeline_lidar.py: time inside the flushes, their number, keys sent to HBM)
#endif
  for (int32_t j = 0; __ballot(k0 + j <= k1) != 0ull; j++) {
    bool spill = false, added = false;
    u64 key = KEY_EMPTY;
    if (k0 + j <= k1) {
      key = pack_key(dd.cur[0], dd.cur[1], dd.cur[2]);
      spill = block_in_workspace(f, dd.cur[0], dd.cur[1], dd.cur[2]) && !lset_insert<LSET>(lset, dd.cur[0], dd.cur[1], dd.cur[2], key, &added);
      dda_step(dd);
    }
    nset += (int32_t)__popcll(__ballot(added));
    if (__ballot(spill)) {
      int4 rec = make_int4(0, 0, 0, 0);
      const bool first = spill && mark_block(m, key, f.frame_id, f.cam_bit, &rec);
      view_append(cnt, view_list, list_cap, first, rec, lane);
    }
    const bool last = __ballot(k0 + j + 1 <= k1) == 0ull;
#ifdef NVBX_WG_TIMES
    const unsigned long long tf0 = (last || nset > LSET_FLUSH) ? wall_clock64() : 0ull;
    if (last || nset > LSET_FLUSH) NVBX_TV(0, 1, tf0);          // (slot 1: start of the LAST flush; slots 3, 4, 5: its phases, flush_set)
#endif
    if (last || nset > LSET_FLUSH) {
      flush_set<LSET, FR, Sensor::kProbeDepth>(m, f, lset, lkeys, cnt, view_list, list_cap, lane, !last);
#ifdef NVBX_WG_TIMES
      t_flush += wall_clock64() - tf0; n_flush++; n_keys += (unsigned long long)nset;
#endif
      nset = 0;
    }
  }
#ifdef NVBX_WG_TIMES
  NVBX_TV(0, 2, t_flush); NVBX_TV(0, 6, (n_flush << 32) | n_keys); NVBX_T(0, 7);
#endif
}

template <typename Img, typename Sensor, int NB>
__global__ __launch_bounds__(Sensor::kThreads) NVBX_MARK_VIEW_ATTR void k_mark_view(DMap m, FrameSet<Img, NB> fs, Sensor sensor, int4* view_list, int32_t list_cap,
                                                                int32_t reset_esdf_dirty, int32_t n_edt_wg, EsdfArgs ea, TraceRiderT<NB> tr) {
  // LDS: the tile's key set (2 * LSET u64), or -- when a distance transform rides (camera, classic order) -- at least an EdtShared; sized by
  // the launch (mark_view_smem below): EVERY workgroup of the launch holds it, the riders too, and it decides how many are resident
  // (a batch of 8 cameras: 2 688 tile workgroups beside 1 200 sphere-tracing ones)
  extern __shared__ __align__(16) unsigned char smem[];
  mark_view_body<Img, Sensor, NB>(m, fs, sensor, view_list, list_cap, reset_esdf_dirty, n_edt_wg, ea, tr, (int32_t)blockIdx.x, smem);
}
// Two mappers' view-marking launches in ONE grid (nvbx_integrate_depth_pair: the background and the foreground mapper of a MultiMapper's dynamic / human
// mapping types take the same depth frame, split by a mask, one after the other -- four dependent launches of 6-8 us each for two small jobs).  A workgroup
// runs mapper a's body or mapper b's, numbered from that mapper's own 0 (n_tiles tiles, then its riders; n_wg in all); nothing is shared between the two maps.
template <typename Img> struct MarkViewArgs { DMap m; FrameSet<Img, 1> fs; int4* view_list; int32_t list_cap, reset_esdf_dirty, n_edt_wg; EsdfArgs ea; TraceRiderT<1> tr; int32_t n_tiles, n_wg; };
template <typename Img>
__global__ __launch_bounds__(CameraSensor::kThreads) void k_mark_view_pair(MarkViewArgs<Img> a, MarkViewArgs<Img> b) {
  extern __shared__ __align__(16) unsigned char smem[];
  // dispatch order [a's tiles][b's tiles][a's riders][b's riders]: the tiles are each map's longest chain (~9 us) and start first (measured against
  // [all of a][all of b]: 11.4 against 11.6 us -- within the noise; kept because it is the order a single mapper's launch has).  Every segment is a
  // multiple of 8 long.
  const int32_t g = (int32_t)blockIdx.x, at = a.n_tiles, bt = b.n_tiles, ar = a.n_wg - a.n_tiles;
  bool is_a; int32_t wg;
  if (g < at) { is_a = true; wg = g; }
  else if (g < at + bt) { is_a = false; wg = g - at; }
  else if (g < at + bt + ar) { is_a = true; wg = at + (g - at - bt); }
  else { is_a = false; wg = bt + (g - at - bt - ar); }
  if (is_a) mark_view_body<Img, CameraSensor, 1>(a.m, a.fs, CameraSensor{}, a.view_list, a.list_cap, a.reset_esdf_dirty, a.n_edt_wg, a.ea, a.tr, wg, smem);
  else mark_view_body<Img, CameraSensor, 1>(b.m, b.fs, CameraSensor{}, b.view_list, b.list_cap, b.reset_esdf_dirty, b.n_edt_wg, b.ea, b.tr, wg, smem);
}
static_assert(2 * sizeof(MarkViewArgs<DepthF32>) <= 4096, "k_mark_view_pair: kernel arguments");

// ------------------------------------------------------------------------------------------------ LiDAR view calculation over a dense "seen in this scan" grid
// A 200 m scan walks its 16 k (sub-sampled) rays through ~10^6 blocks to find the ~112 k distinct ones.  k_mark_view<Lidar> above decides "first
// ray through this block?" with one compare-and-swap per key on the hash entry's stamp; a first version of this path decided it with one returning
// atomicOr per key on a bit of a dense grid.  Both take ~58 us, and the per-bundle time stamps (tools/wg_timeline_lidar_grid.py) say why: a bundle
// with ONE far key waits 30-40 us for its atomic like a bundle with 500 -- ~10^5 returning atomics on scattered addresses are served at ~3 G/s by the
// memory side, whoever issues them.  So the marking launch issues NO atomic at all:
//   k_mark_view_grid : the walk (same code as k_mark_view: view_ray_setup, dda_jump, dda_step); a visited block is a plain STORE of 1 to its byte of a
//                      dense grid around the sensor (idempotent: any number of rays may visit) + a store of 1 to the byte of its 4 x 4 x 4 cell in
//                      a coarse map.  The grid is cell-major -- the 64 bytes of a cell are one 64-B line.  (Every visit stores: ~700 k redundant byte
//                      stores around the sensor cost less than looking first -- VG_NEAR > 0 builds the look-before-store variant, measured slower.)
//   k_scan_view_grid : reads the coarse map (0.8 MB for a 200 m box at 0.8 m blocks), the lines of the touched cells, and appends {tag, x, y, z} per
//                      set byte to the view list -- one reservation per wavefront -- and puts every byte it found back to 0: the grid is all-zero
//                      again when the scan's launches are done.
//   k_resolve_view   : one lane per tagged record: hash lookup or insert (the wavefront's new blocks pop their slots together), entry stamp, slot
//                      written into the record.
// Blocks outside the box (none, if the box was sized from the sensor's range: a ray then ends inside by construction) take mark_block directly.
// Same block set as k_mark_view<Lidar>; only the de-duplication differs.
struct ViewGrid {
  uint8_t* fine;           // byte cell * 64 + (lx & 3) + 4 (ly & 3) + 16 (lz & 3), cell = ((lz >> 2) * ncy + (ly >> 2)) * ncx + (lx >> 2), l = block - o
  uint8_t* coarse;         // byte per cell (padded to a multiple of 4)
  int32_t ox, oy, oz;      // block index of the box's minimum corner
  int32_t ncx, ncy, ncz;   // cells per axis (<= 256: local block coordinates are 10 bits)
  int32_t cx, cy, cz;      // the sensor's block
  uint32_t tag;            // slot field of a record waiting for k_resolve_view: 0x80000000 | view frame id (never a slot: capacity <= 2^24)
};
#ifndef NVBX_VG_NEAR
#define NVBX_VG_NEAR 0              // (0: every visit stores.  24 / 48: 19.3 / 19.9 us for the launch instead of 16.6 -- the stores were never what waited)
#define NVBX_VG_CHUNK 16
#endif
constexpr int VG_NEAR = NVBX_VG_NEAR, VG_CHUNK = NVBX_VG_CHUNK;
constexpr uint32_t VG_NONE = 0xFFFFFFFFu;

#ifndef NVBX_VIEW_GRID_ATTR
#define NVBX_VIEW_GRID_ATTR
#endif
template <typename Img>
__global__ __launch_bounds__(64) NVBX_VIEW_GRID_ATTR void k_mark_view_grid(DMap m, FrameSet<Img, 1> fs, LidarSensor sensor, int4* view_list, int32_t list_cap,
                                                       int32_t reset_esdf_dirty, int32_t fence_report, ViewGrid vg) {
  constexpr int TR = LidarSensor::kTileRows, TC = LidarSensor::kTileCols, NSEG = LidarSensor::kSegments, C = VG_CHUNK;
  static_assert(TR * TC * NSEG <= 64, "one wavefront per bundle of rays");
  const int lane = (int)threadIdx.x;
  const Frame& f = fs.f[0];
  const Img& depth = fs.img[0];
  NVBX_T(0, 0);
  if (blockIdx.x == 0 && lane == 0) __hip_atomic_store(&m.host_mirror[4], fence_report, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);    // (k_mark_view: frames.hip's fence)
  // XCD-aware numbering, as k_mark_view: the bundles of one XCD are a contiguous band of ray rows
  const int tiles_x = (f.n_ray_cols + TC - 1) / TC, tiles_y = (f.n_ray_rows + TR - 1) / TR;
  const int n_tiles = tiles_x * tiles_y, per_xcd = (n_tiles + NSH - 1) / NSH;
  const int wg = (int)blockIdx.x;
  const int tile = (wg & (NSH - 1)) * per_xcd + (wg >> 3);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int ray = lane / NSEG, seg = lane % NSEG;
  const int ri = ty * TR + ray / TC, ci = tx * TC + ray % TC;
  bool active = (wg >> 3) < per_xcd && tile < n_tiles && ray < TR * TC && ri < f.n_ray_rows && ci < f.n_ray_cols;
  int prow = ri * f.subsample; if (prow >= f.rows) prow = f.rows - 1;
  int pcol = ci * f.subsample; if (pcol >= f.cols) pcol = f.cols - 1;
  const float d = active ? depth(pix(prow, pcol, f.cols)) : 0.0f;
  if (wg == 0 && lane == 0) m.counters[C_VIEW_COUNT + ((f.frame_id + 1) & 3)] = 0;   // next frame's counter
  if (wg == 0 && lane < NSH) { *shc_at(m, S_LIDAR_SPARSE, lane, 0) = 0; *shc_at(m, S_LIDAR_SPARSE, lane, 1) = 0; }
  if (reset_esdf_dirty && wg == 0 && lane < NSH) *shc_at(m, S_LIST_ESDF_DIRTY, lane, 0) = 0;
  Dda dd{};
  float inv_dt[3] = {0.0f, 0.0f, 0.0f};
  const int32_t nsteps = view_ray_setup(f, sensor, active, d, prow, pcol, dd, inv_dt);
  NVBX_TV(0, 1, wall_clock64() + (unsigned long long)(nsteps & 0));       // (the depth pixel has arrived, the ray is set up)
  int32_t k0 = 0, k1 = nsteps;
  if (NSEG > 1 && nsteps >= 0) {
    const int32_t q = (nsteps + NSEG) / NSEG;
    k0 = seg * q; k1 = min(nsteps, k0 + q - 1);
    if (k0 > nsteps) k1 = -1;
    else if (k0 > 0) dda_jump(dd, k0, inv_dt);
  }
  NVBX_TV(0, 2, wall_clock64() + (unsigned long long)(dd.cur[0] & 0));    // (this lane stands at the start of its segment)
  int32_t* cnt = &m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  const uint32_t NX = 4u * (uint32_t)vg.ncx, NY = 4u * (uint32_t)vg.ncy, NZ = 4u * (uint32_t)vg.ncz;
  for (int32_t base = k0; __ballot(base <= k1) != 0ull; base += C) {
    uint32_t code[VG_NEAR > 0 ? C : 1], w[VG_NEAR > 0 ? C : 1];
#pragma unroll
    for (int i = 0; i < C; i++) {
      if (VG_NEAR > 0) code[i] = VG_NONE;
      bool spill = false; u64 key = KEY_EMPTY;
      if (base + i <= k1) {
        const int32_t bx = dd.cur[0], by = dd.cur[1], bz = dd.cur[2];
        if (block_in_workspace(f, bx, by, bz)) {
          const uint32_t lx = (uint32_t)(bx - vg.ox), ly = (uint32_t)(by - vg.oy), lz = (uint32_t)(bz - vg.oz);
          if (lx < NX && ly < NY && lz < NZ) {
            const uint32_t cell = ((lz >> 2) * (uint32_t)vg.ncy + (ly >> 2)) * (uint32_t)vg.ncx + (lx >> 2);
            const uint32_t at = cell * 64u + ((lx & 3u) | ((ly & 3u) << 2) | ((lz & 3u) << 4));
            const uint32_t nx = (uint32_t)(bx - vg.cx + VG_NEAR), ny = (uint32_t)(by - vg.cy + VG_NEAR), nz = (uint32_t)(bz - vg.cz + VG_NEAR);
            if (VG_NEAR > 0 && nx < 2u * VG_NEAR && ny < 2u * VG_NEAR && nz < 2u * VG_NEAR) code[VG_NEAR > 0 ? i : 0] = at;      // (looked at first, below)
            else { vg.fine[at] = 1; vg.coarse[cell] = 1; }
          } else { spill = true; key = pack_key(bx, by, bz); }
        }
        dda_step(dd);
      }
      if (__ballot(spill)) {                     // outside the box (a box sized from the sensor's range holds every ray): the hash decides
        int4 rec = make_int4(0, 0, 0, 0);
        const bool first = spill && mark_block(m, key, f.frame_id, f.cam_bit, &rec);
        view_append(cnt, view_list, list_cap, first, rec, lane);
      }
    }
    if (base == k0) NVBX_TV(0, 3, wall_clock64());
    if (VG_NEAR > 0) {
      // near the sensor: the chunk's bytes are loaded together and stored only where they read 0 (a stale 0 costs a store, nothing else)
#pragma unroll
      for (int i = 0; i < C; i++) { w[i] = 1u; if (code[i] != VG_NONE) w[i] = (uint32_t)vg.fine[code[i]]; }
#pragma unroll
      for (int i = 0; i < C; i++) if (code[i] != VG_NONE && !w[i]) { vg.fine[code[i]] = 1; vg.coarse[code[i] >> 6] = 1; }
      if (base == k0) NVBX_TV(0, 4, wall_clock64() + (unsigned long long)(w[0] & 0u));
    }
  }
  NVBX_T(0, 7);
}

// Up to K keys per lane -> pool slots, the dependent round trips taken together: the first PD probe positions of every key, then the inserts of
// the blocks that are new (compare-and-swap on the entries; the wavefront's winners pop their slots with ONE atomicSub on the free-stack top and
// one atomicMax on the high-water mark -- flush_set's B'), then whatever is left (a longer probe chain, a lost insert) by hash_insert.  The caller
// OWNS these keys for the launch (nobody else looks them up or stamps them), so the entry stamp is a plain store.  Whole wavefront must call.
template <int K, int PD, bool STAMP = true>
__device__ inline void resolve_keys(const DMap& m, const u64 (&key)[K], const bool (&valid)[K], uint32_t want, uint32_t (&slot)[K], int lane) {
  uint32_t h[K]; uint4 e[K][PD];
#pragma unroll
  for (int k = 0; k < K; k++) { int32_t x, y, z; unpack_key(key[k], &x, &y, &z); h[k] = valid[k] ? table_pos(m, x, y, z) : 0u; }
#pragma unroll
  for (int k = 0; k < K; k++) if (valid[k]) {
#pragma unroll
    for (int q = 0; q < PD; q++) e[k][q] = ld_entry(m, (h[k] + q) & m.mask);
  }
  bool done[K], ins[K], won[K]; uint32_t hpos[K];
  bool any_ins = false;
#pragma unroll
  for (int k = 0; k < K; k++) {
    done[k] = false; ins[k] = false; won[k] = false; hpos[k] = 0u; slot[k] = SLOT_NONE;
    if (valid[k]) {
      bool open = true; int qe = -1;
#pragma unroll
      for (int q = 0; q < PD; q++) {
        const u64 kq = ((u64)e[k][q].y << 32) | (u64)e[k][q].x;
        if (open && !done[k] && kq == key[k]) { slot[k] = e[k][q].z; hpos[k] = (h[k] + q) & m.mask; done[k] = true; }
        if (open && kq == KEY_EMPTY) { open = false; if (!done[k]) qe = q; }
      }
      if (done[k] && slot[k] == SLOT_INVALID) done[k] = false;            // (being inserted by somebody else right now: hash_insert below waits)
      else if (!done[k] && qe >= 0) { ins[k] = true; any_ins = true; hpos[k] = (h[k] + qe) & m.mask; }
    }
  }
  if (__ballot(any_ins)) {
    u64 oldk[K];
#pragma unroll
    for (int k = 0; k < K; k++) if (ins[k]) oldk[k] = atomicCAS(&m.table[hpos[k]].key, KEY_EMPTY, key[k]);
    int32_t wtotal = 0, wpre[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      won[k] = ins[k] && oldk[k] == KEY_EMPTY;
      const u64 mask = __ballot(won[k]);
      wpre[k] = wtotal + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
      wtotal += (int32_t)__popcll(mask);
    }
    if (wtotal) {
      int32_t top = 0;
      if (lane == 0) {
        top = atomicSub(&m.counters[C_FREE_TOP], wtotal);
        if (top < wtotal) { atomicAdd(&m.counters[C_FREE_TOP], wtotal - (top > 0 ? top : 0)); atomicExch(&m.counters[C_OVERFLOW], 1); }     // pool exhausted: give back what was not there
      }
      top = __shfl(top, 0);
      int32_t hwm = 0;
#pragma unroll
      for (int k = 0; k < K; k++) if (won[k]) {
        const int32_t idx = top - 1 - wpre[k];
        slot[k] = idx >= 0 ? m.free_stack[idx] : SLOT_NONE;
        if (slot_ok(slot[k])) {
          int32_t x, y, z; unpack_key(key[k], &x, &y, &z);
          m.slot_index[3 * slot[k]] = x; m.slot_index[3 * slot[k] + 1] = y; m.slot_index[3 * slot[k] + 2] = z;
          m.slot_entry[slot[k]] = hpos[k];
          atomicOr(&m.slot_flags[slot[k]], F_TSDF);
          hwm = max(hwm, (int32_t)slot[k] + 1);
        }
        __hip_atomic_store(&m.table[hpos[k]].slot, slot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        done[k] = true;
      }
#pragma unroll
      for (int o = 32; o; o >>= 1) hwm = max(hwm, __shfl_xor(hwm, o));
      if (lane == 0 && hwm) atomicMax(&m.counters[C_HIGH_WATER], hwm);
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) if (valid[k]) {
    if (!done[k]) {
      int32_t x, y, z; unpack_key(key[k], &x, &y, &z);
      bool is_new;
      const int32_t hi = hash_insert(m, x, y, z, F_TSDF, &is_new);
      if (hi < 0) { slot[k] = SLOT_NONE; continue; }
      hpos[k] = (uint32_t)hi;
      uint32_t s = SLOT_INVALID;
      while (s == SLOT_INVALID) s = ld_slot_acquire(&m.table[hi]);
      slot[k] = s;
    }
    if (STAMP) m.table[hpos[k]].stamp = want;
  }
}

// The launches behind k_mark_view_grid.  k_scan_view_grid: a wavefront takes four 64-B lines of the coarse map (256 cells; the four lines from
// four far-apart places: the cells around the sensor are all touched and lie in a few hundred neighbouring lines -- taken as consecutive lines they
// gave a few wavefronts 60 cells each and the launch 28 us), lists the touched cells in LDS, reads their lines (four lanes x 16 bytes per cell,
// sixteen cells per round) once to count and once more -- from the L2 -- to write {tag, x, y, z} per set byte behind ONE reservation; whatever it
// found set goes back to 0.
__device__ inline int32_t vg_nonzero_bytes(uint32_t b) { return (int32_t)((b & 0xFFu) != 0u) + (int32_t)((b & 0xFF00u) != 0u) + (int32_t)((b & 0xFF0000u) != 0u) + (int32_t)((b >> 24) != 0u); }
constexpr int VG_SCAN_WAVES = 8;          // wavefronts per scanning workgroup: ONE reservation per workgroup (3 000 per-wavefront reservations on the
                                          // view counter were most of a 15 us launch: returning atomics on one address are served one after the other)
__global__ __launch_bounds__(64 * VG_SCAN_WAVES) void k_scan_view_grid(DMap m, uint32_t frame_id, int4* view_list, int32_t list_cap, ViewGrid vg) {
  __shared__ uint32_t s_cells[VG_SCAN_WAVES][256];
  __shared__ int32_t s_total[VG_SCAN_WAVES], s_base;
  int32_t* cnt = &m.counters[C_VIEW_COUNT + (frame_id & 3)];
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int32_t n_cells = vg.ncx * vg.ncy * vg.ncz, n_words = (n_cells + 3) >> 2, n_lines = (n_words + 15) >> 4;
  const int32_t n_waves = (int32_t)gridDim.x * VG_SCAN_WAVES, me = (int32_t)blockIdx.x * VG_SCAN_WAVES + wave;       // (4 n_waves >= n_lines: one pass)
  uint32_t* cw = reinterpret_cast<uint32_t*>(vg.coarse);
  uint4* fq = reinterpret_cast<uint4*>(vg.fine);
  uint32_t* cells = s_cells[wave];
  const int32_t line = me + (lane >> 4) * n_waves;
  const int32_t i = line * 16 + (lane & 15);
  const uint32_t v = (line < n_lines && i < n_words) ? cw[i] : 0u;
  if (v) cw[i] = 0u;
  int32_t n = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool t = ((v >> (8 * k)) & 0xFFu) != 0u;
    const u64 mask = __ballot(t);
    if (t) cells[n + (int32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)(4 * i + k);
    n += (int32_t)__popcll(mask);
  }
  const int sub = lane >> 2, part = lane & 3;             // sixteen cells per round, four lanes (16 bytes each) per cell
  int32_t mine = 0;
  for (int32_t it = 0; it < n; it += 16) {
    uint4 b = make_uint4(0u, 0u, 0u, 0u);
    if (it + sub < n) b = fq[(size_t)cells[it + sub] * 4 + part];
    mine += vg_nonzero_bytes(b.x) + vg_nonzero_bytes(b.y) + vg_nonzero_bytes(b.z) + vg_nonzero_bytes(b.w);
  }
  int32_t total = mine;
#pragma unroll
  for (int o = 32; o; o >>= 1) total += __shfl_xor(total, o);
  if (lane == 0) s_total[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t all = 0;
#pragma unroll
    for (int w = 0; w < VG_SCAN_WAVES; w++) all += s_total[w];
    s_base = all ? atomicAdd(cnt, all) : 0;
  }
  __syncthreads();
  if (!total) return;
  int32_t pos = s_base;
#pragma unroll
  for (int w = 0; w < VG_SCAN_WAVES; w++) if (w < wave) pos += s_total[w];
  for (int32_t it = 0; it < n; it += 16) {
    const bool have = it + sub < n;
    const uint32_t cell = have ? cells[it + sub] : 0u;
    uint4 b = make_uint4(0u, 0u, 0u, 0u);
    if (have) b = fq[(size_t)cell * 4 + part];
    if (b.x | b.y | b.z | b.w) fq[(size_t)cell * 4 + part] = make_uint4(0u, 0u, 0u, 0u);
    const int32_t cxi = (int32_t)(cell % (uint32_t)vg.ncx), cyi = (int32_t)((cell / (uint32_t)vg.ncx) % (uint32_t)vg.ncy), czi = (int32_t)(cell / ((uint32_t)vg.ncx * (uint32_t)vg.ncy));
    const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (__ballot(bw[q] != 0u) == 0ull) continue;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool t = ((bw[q] >> (8 * k)) & 0xFFu) != 0u;
        const u64 mask = __ballot(t);
        if (t) {
          const int32_t at = pos + (int32_t)__popcll(mask & ((1ull << lane) - 1ull));
          const int j = part * 16 + q * 4 + k;               // byte of the cell: lx & 3 | (ly & 3) << 2 | (lz & 3) << 4
          if (at < list_cap) view_list[at] = make_int4((int32_t)vg.tag, vg.ox + 4 * cxi + (j & 3), vg.oy + 4 * cyi + ((j >> 2) & 3), vg.oz + 4 * czi + (j >> 4));
        }
        pos += (int32_t)__popcll(mask);
      }
    }
  }
}
// k_resolve_view: the records the scan left tagged, one per lane -- the slot replaces the tag (records of blocks outside the box carry their slot already)
__global__ __launch_bounds__(256) void k_resolve_view(DMap m, uint32_t frame_id, int4* view_list, int32_t list_cap, uint32_t tag) {
  const int32_t n = min(m.counters[C_VIEW_COUNT + (frame_id & 3)], list_cap);
  const uint32_t want = (frame_id << 8) | 1u;
  const int lane = (int)(threadIdx.x & 63);
  for (int32_t i0 = (int32_t)blockIdx.x * 256 + (int32_t)(threadIdx.x & ~63u); i0 < n; i0 += (int32_t)gridDim.x * 256) {
    const int32_t i = i0 + lane;
    int4 rec = make_int4(0, 0, 0, 0);
    if (i < n) rec = view_list[i];
    u64 key[1]; bool valid[1]; uint32_t slot[1];
    valid[0] = i < n && (uint32_t)rec.x == tag;
    key[0] = pack_key(rec.y, rec.z, rec.w);
    if (__ballot(valid[0]) == 0ull) continue;
    // (no entry stamp: Entry::stamp de-duplicates the tiles of a CAMERA frame and carries a batch's camera masks; a scan's view is its view list --
    //  nothing reads the stamp of a LiDAR frame, and 112 k scattered 4-byte stores are 112 k lines written back)
    resolve_keys<1, 2, false>(m, key, valid, want, slot, lane);
    if (valid[0]) view_list[i].x = (int32_t)slot[0];
  }
}

// A wave-uniform value the inner loop uses as a VALU operand, parked in a vector register: the LiDAR instantiation needs ~100 scalar
// registers, the 8-waves-per-SIMD budget leaves 72, and every use of a spilled one is a v_readlane in the per-voxel path.
template <typename T> __device__ inline T in_vgpr(T x) { asm volatile("" : "+v"(x)); return x; }

// Dependent-access chain: {view count, view record} -> {depth gather, voxel} -> store.  The record of the first block
// is fetched speculatively beside the count, the voxel is fetched before the projection decides whether it is needed,
// and the flag / dirty-list atomics of lane 0 are issued first and consumed last.
// Occupancy: 8 waves per SIMD (four of these workgroups per CU) is asked for explicitly -- the kernel's ~100 scalar registers would
// otherwise cost the eighth wave, and on the LiDAR map (112 k blocks per scan) residency is throughput; the grid is exactly the 1024
// workgroups that are then resident together (tools/integ_grid_sweep.sh: 768 / 1024 workgroups at 6 waves 499 / 569 us, 1024 at 8 waves 430-470).
// Plain = every camera of the launch is frame_is_plain() (nvbx_internal.h): the occupancy / weighting-mode / decay switches fold away
// at compile time -- same arithmetic, 12 % fewer issue cycles on the LiDAR launch (tools/variant_ab.sh).
template <typename Img, typename Sensor, int NB, bool Plain>
__device__ inline void integrate_tsdf_worker(const DMap& m, const FrameSet<Img, NB>& fs, const Sensor& sensor, const int4* view_list, int32_t list_cap,
                                             int32_t mesh_list, int32_t* view_export, int32_t view_export_cap, int32_t spec_lanes, const uint8_t* view_class,
                                             const int32_t wgi, const int32_t n_wg, const int32_t* dense_list = nullptr) {
  const Frame& f0 = fs.f[0];
  const int tid = threadIdx.x, lane = tid & 63;
  // The view records are taken 64 at a time: lane j of every wavefront fetches the record of the j-th block this workgroup will
  // process next (speculatively, beside the count) and transforms that block's origin into the sensor frame; the block loop then
  // reads slot and origin out of lane j (v_readlane: scalar operands from there on).  The record fetch leaves the per-block
  // dependent chain and the 3 x 3 transform is paid once per block and wavefront instead of once per voxel.
  // Only the first `spec_lanes` lanes fetch speculatively (a host hint: the view count the GPU last reported / the grid, + 1): a camera
  // frame has ~300 blocks in view and about as many workgroups, so one record per wavefront is wanted -- 64 speculative 16-B records from
  // each of 8 x 1024 wavefronts were 8 MB of HBM traffic for 3.5 MB of work (PMC, profiles/r02z_pmc.json).  A lane the hint left out
  // fetches once the count is known (a dependent load, only when the view grew by more than the hint's margin).
  // which records this workgroup takes: runs of consecutive records (neighbouring blocks: the same patch of the depth image) stay on one XCD's L2
  // (xcd_chunked, nvbx_internal.h); the LiDAR work list is dealt out per shard already
  const int32_t wgr = dense_list ? wgi : xcd_chunked(wgi, n_wg);
  int32_t mine = wgr + lane * n_wg;
  int4 rec = (!dense_list && lane < spec_lanes && mine < list_cap) ? view_list[mine] : make_int4((int32_t)SLOT_NONE, 0, 0, 0);
  int32_t n = m.counters[C_VIEW_COUNT + (f0.frame_id & 3)];
  if (n > list_cap) n = list_cap;
  if (wgi == 0 && tid == 192) __hip_atomic_store(&m.host_mirror[2], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // next launch's hint (the VIEW count)
  // LiDAR behind the beam-centric launch: that launch has left the records it did NOT take as a work list (indices into the view list, one
  // region per shard) -- this launch deals THOSE out, so every workgroup gets the same number of blocks.  Dealing out the whole view list and
  // skipping the taken records (60 %) left a workgroup with Binomial(110, 0.4) blocks: 44 +- 5, the slowest of 1024 with ~60.
  int32_t dpre[NSH + 1];
  auto dense_at = [&](int32_t k) -> int32_t {          // (selects only: a dynamically indexed dpre[] would live in scratch memory)
    int sh = 0; int32_t base = 0;
#pragma unroll
    for (int q = 1; q < NSH; q++) if (k >= dpre[q]) { sh = q; base = dpre[q]; }
    return dense_list[(size_t)sh * list_cap + (k - base)];
  };
  if (dense_list) {
    int32_t c[NSH];
#pragma unroll
    for (int q = 0; q < NSH; q++) c[q] = *shc_at(m, S_LIDAR_SPARSE, q, 1);
    dpre[0] = 0;
#pragma unroll
    for (int q = 0; q < NSH; q++) dpre[q + 1] = dpre[q] + min(c[q], list_cap);
    n = dpre[NSH];
    if (mine < n) rec = view_list[dense_at(mine)];
  } else {
    if (lane >= spec_lanes && mine < n) rec = view_list[mine];
    // (LiDAR without the work list: blocks the beam-centric launch k_lidar_sparse has already updated are skipped -- their record reads as "no slot")
    if (view_class && mine < n && view_class[mine]) rec.x = (int32_t)SLOT_NONE;
  }
  // nvbx_set_view_export: the frame's block indices also go to a caller-owned packed buffer [1 + cap][3] (row 0 = count) --
  // the message of the multi-GPU exchange, written here instead of by an export launch
  if (view_export && wgi == 0 && tid == 0) { view_export[0] = min(n, view_export_cap); view_export[1] = 0; view_export[2] = 0; }
  // pool growth: the free-slot count after this frame's allocations goes to pinned host memory (not waited for)
  if (wgi == 0 && tid == 64) __hip_atomic_store(&m.host_mirror[0], m.counters[C_FREE_TOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (wgi == 0 && tid == 128) __hip_atomic_store(&m.host_mirror[1], m.counters[C_HIGH_WATER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  float off0[3] = {0.0f, 0.0f, 0.0f};
  if (NB == 1) sensor_voxel_offset(f0, vx, vy, vz, off0);          // (a batch rotates the offset per camera inside the loop)
  // LiDAR: the image geometry the four-tap gather needs per voxel lives in vector registers (see in_vgpr)
  Frame fl = f0; Img img0 = fs.img[0];
  if (Sensor::kLongRays && NB == 1) { fl.cols = in_vgpr(f0.cols); fl.rows = in_vgpr(f0.rows); img0.p = in_vgpr(fs.img[0].p); }
  for (int32_t i0 = wgr; i0 < n; i0 += 64 * n_wg) {
    if (i0 != wgr) {
      mine = i0 + lane * n_wg;
      if (dense_list) rec = mine < n ? view_list[dense_at(mine)] : make_int4((int32_t)SLOT_NONE, 0, 0, 0);
      else {
        rec = mine < n ? view_list[mine] : make_int4((int32_t)SLOT_NONE, 0, 0, 0);
        if (view_class && mine < n && view_class[mine]) rec.x = (int32_t)SLOT_NONE;
      }
    }
    if (view_export && tid < 64 && mine < n && mine < view_export_cap) { int32_t* e = view_export + 3 * (1 + (int64_t)mine); e[0] = rec.y; e[1] = rec.z; e[2] = rec.w; }
    float org[3] = {0.0f, 0.0f, 0.0f};
    if (NB == 1) sensor_block_origin(f0, rec.y, rec.z, rec.w, org);
#pragma unroll 1
    for (int32_t j = 0; j < 64 && i0 + j * n_wg < n; j++) {               // (uniform)
      const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane(rec.x, j);               // pool slot (stable across hash rebuilds)
      if (!slot_ok(slot)) continue;
      float2* vp = &m.tsdf[(size_t)slot * 512 + tid];
      const float2 cur_c = *vp;
      // batch: which cameras had this block in view = the mask k_mark_view left in the entry's stamp (uniform per block)
      uint32_t cams = 1u;
      if (NB > 1) cams = __builtin_amdgcn_readfirstlane(m.table[m.slot_entry[slot]].stamp & 0xFFu);
      uint32_t old = 0;
      if (tid == 0) old = atomicOr(&m.slot_flags[slot], F_TSDF | F_DIRTY_ESDF | F_DIRTY_MESH | ((Sensor::kLongRays && (Plain || !f0.occupancy)) ? F_BAND_STALE : 0u));
      if (!Sensor::kLongRays && tid == 64) m.slot_cam[slot] = (f0.frame_id << 8) | cams;       // the camera view decayTsdfExcludeLastView<Camera> spares
      const int32_t bx = __builtin_amdgcn_readlane(rec.y, j), by = __builtin_amdgcn_readlane(rec.z, j), bz = __builtin_amdgcn_readlane(rec.w, j);
      float2 fin = cur_c;            // the voxel as this launch leaves it: the cameras' updates applied in order, exactly as separate calls would
      bool touched = false;
#pragma unroll 1
      for (int c = 0; c < (NB > 1 ? fs.n : 1); c++) {
        if (NB > 1 && !((cams >> c) & 1u)) continue;       // uniform
        const Frame& f = (Sensor::kLongRays && NB == 1) ? fl : fs.f[c];
        const Img& img = (Sensor::kLongRays && NB == 1) ? img0 : fs.img[c];
        float pc[3];
        if (NB == 1) {
          pc[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org[0]), j)) + off0[0];
          pc[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org[1]), j)) + off0[1];
          pc[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org[2]), j)) + off0[2];
        } else {
          float o[3], d[3];
          sensor_block_origin(f, bx, by, bz, o); sensor_voxel_offset(f, vx, vy, vz, d);
          pc[0] = o[0] + d[0]; pc[1] = o[1] + d[1]; pc[2] = o[2] + d[2];
        }
        float ds = 0.0f, vd = 0.0f;
        const int got = sensor.sample(f, img, pc, &ds, &vd);
        if (Plain) {
          if (got > 0 && tsdf_fuse_plain(f, &fin, ds, vd)) touched = true;
        } else if (f.occupancy) {     // occupancy mapper: the pool holds log-odds (nvbx_internal.h occupancy_update)
          if (got > 0) { fin = make_float2(occupancy_update(f, fin.x, ds, vd), 0.0f); touched = true; }
        } else {
          if (got < 0 && f.invalid_decay >= 0.0f) { fin = make_float2(fin.x, fin.y * f.invalid_decay); touched = true; }
          if (got > 0 && tsdf_fuse(f, &fin, ds, vd)) touched = true;
        }
      }
      if (touched) *vp = fin;
      if (Plain || !f0.occupancy) {
        // band vote for the colour integrator (F_BAND, nvbx_internal.h): exact, so set AND cleared here, one bit per wavefront
        if (!Sensor::kLongRays) {      // (LiDAR: marked stale above instead)
          // one workgroup per block and a few hundred blocks: a block-wide vote and ONE atomic are cheaper here than a bit per wavefront
          const int any_band = __syncthreads_or(in_band(fin.x, fin.y, f0.trunc) ? 1 : 0);
          if (tid == 0) { if (any_band) atomicOr(&m.slot_flags[slot], F_BAND); else atomicAnd(&m.slot_flags[slot], ~F_BAND); if (old & F_BAND_STALE) atomicAnd(&m.slot_flags[slot], ~F_BAND_STALE); }
        }
      }
      if (tid == 0) {
        if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)slot);
        if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, (int32_t)slot);
      }
    }
  }
}
template <typename Img, typename Sensor, int NB, bool Plain>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_integrate_tsdf(DMap m, FrameSet<Img, NB> fs, Sensor sensor, const int4* view_list, int32_t list_cap,
                                                        int32_t mesh_list, int32_t* view_export, int32_t view_export_cap, int32_t spec_lanes, const uint8_t* view_class,
                                                        const int32_t* dense_list) {
  NVBX_INV_WRITER_BEGIN(m);
  integrate_tsdf_worker<Img, Sensor, NB, Plain>(m, fs, sensor, view_list, list_cap, mesh_list, view_export, view_export_cap, spec_lanes, view_class, (int32_t)blockIdx.x, (int32_t)gridDim.x, dense_list);
  NVBX_INV_WRITER_END(m);
}

// Pipelined order, fused: TSDF update of frame i + 1, colour integration of frame i (from the candidate records the riders
// of the view-marking launch discovered) and the distance transform of ESDF update i in ONE launch -- two launches per frame:
//   k_mark_view            view marking (i+1) || sphere tracing (i) || colour candidates (i) || ESDF site marking (i)
//   k_integrate_tsdf_color TSDF update (i+1)  || colour integration (i)                      || distance transform (i)
// What makes the three parts of this launch independent: the colour workers read no block flag (their candidates were fixed before the
// launch: the TSDF update rewrites F_BAND beside them) and no TSDF voxel; the distance transform reads the site masks of a marking pass that
// has finished (previous launch) and writes ESDF voxels only; the update appends to an ESDF-dirty list that pass has emptied.  Every part
// reads exactly the state separate calls would have shown it.
// Workgroups: [distance transform (512 threads = 8 wavefronts per ESDF block)][TSDF update][colour].
// (occupancy: the compiler's own register choice gives 3 of these 8-wavefront workgroups per CU -- 768 resident of the ~1 080 a frame launches, so the
//  colour workgroups, dispatched last, start late; asking for four per CU with amdgpu_waves_per_eu(8, 8) was measured in round 4: the colour part
//  starts at once but every part runs slower, the launch 9.2 -> 10.0 us -- left at the compiler's choice)
template <typename Img, typename Pix, int NB, bool Plain>
__device__ __forceinline__ void integrate_tsdf_color_body(const DMap& m, const FrameSet<Img, NB>& fs, const CameraSensor& sensor, const int4* view_list, int32_t list_cap,
                                                          int32_t mesh_list, int32_t* view_export, int32_t view_export_cap, int32_t spec_lanes, int32_t n_tsdf_wg,
                                                          const FrameSetC<Pix, NB>& fsc, const float* synth, int32_t srows, int32_t scols, const int4* cand, int32_t cand_cnt_idx,
                                                          int32_t n_edt_wg, const EsdfArgs& ea, const ImportArgs& imp, const int32_t b, const int32_t n_wg_total, unsigned char* smem) {
  NVBX_T(1, 0);
  if (b < n_edt_wg) { esdf_edt_worker<512>(m, ea, (int)b, n_edt_wg, reinterpret_cast<EdtShared*>(smem)); NVBX_T(1, 1); NVBX_T(1, 7); return; }
  if (b < n_edt_wg + n_tsdf_wg) {
    NVBX_INV_WRITER_BEGIN(m);
    integrate_tsdf_worker<Img, CameraSensor, NB, Plain>(m, fs, sensor, view_list, list_cap, mesh_list, view_export, view_export_cap, spec_lanes, nullptr, b - n_edt_wg, n_tsdf_wg);
    NVBX_INV_WRITER_END(m);
    NVBX_T(1, 2); NVBX_T(1, 7);
    return;
  }
  // (multi-GPU, imp.n_wg > 0: the LAST workgroups resolve the peers' gathered block lists into ESDF-dirty flags, nvbx_esdf_mark.h)
  const int32_t n_color_wg = n_wg_total - n_edt_wg - n_tsdf_wg - imp.n_wg;
  if (b >= n_edt_wg + n_tsdf_wg + n_color_wg) { esdf_import_dirty_worker(m, imp, (int64_t)(b - n_edt_wg - n_tsdf_wg - n_color_wg) * 512 + threadIdx.x, (int64_t)imp.n_wg * 512); return; }
  color_integrate_list_worker<Pix, NB>(m, fsc, synth, srows, scols, mesh_list, cand, cand_cnt_idx, b - n_edt_wg - n_tsdf_wg, n_color_wg);
  NVBX_T(1, 3); NVBX_T(1, 7);
}
template <typename Img, typename Pix, int NB, bool Plain>
__global__ __launch_bounds__(512) NVBX_FUSED_ATTR void k_integrate_tsdf_color(DMap m, FrameSet<Img, NB> fs, CameraSensor sensor, const int4* view_list, int32_t list_cap,
                                                              int32_t mesh_list, int32_t* view_export, int32_t view_export_cap, int32_t spec_lanes, int32_t n_tsdf_wg,
                                                              FrameSetC<Pix, NB> fsc, const float* synth, int32_t srows, int32_t scols, const int4* cand, int32_t cand_cnt_idx,
                                                              int32_t n_edt_wg, EsdfArgs ea, ImportArgs imp) {
  __shared__ __align__(16) unsigned char smem[sizeof(EdtShared)];
  integrate_tsdf_color_body<Img, Pix, NB, Plain>(m, fs, sensor, view_list, list_cap, mesh_list, view_export, view_export_cap, spec_lanes, n_tsdf_wg, fsc, synth, srows, scols, cand, cand_cnt_idx,
                                                 n_edt_wg, ea, imp, (int32_t)blockIdx.x, (int32_t)gridDim.x, smem);
}
// ... and two mappers' fused TSDF-update launches in one grid (nvbx_integrate_depth_pair, see k_mark_view_pair): mapper a's [distance transform][TSDF
// update][colour][union step] workgroups, then mapper b's, each numbered from its own 0.  The general (not "plain") arithmetic path for both.
template <typename Img, typename Pix> struct FusedArgs {
  DMap m; FrameSet<Img, 1> fs; const int4* view_list; int32_t list_cap, mesh_list; int32_t* view_export; int32_t view_export_cap, spec_lanes, n_tsdf_wg;
  FrameSetC<Pix, 1> fsc; const float* synth; int32_t srows, scols; const int4* cand; int32_t cand_cnt_idx, n_edt_wg; EsdfArgs ea; ImportArgs imp; int32_t n_wg; };
template <typename Img, typename Pix>
__global__ __launch_bounds__(512) void k_integrate_tsdf_color_pair(FusedArgs<Img, Pix> a, FusedArgs<Img, Pix> b) {
  __shared__ __align__(16) unsigned char smem[sizeof(EdtShared)];
  // dispatch order: mapper a's workgroups, then mapper b's (dealing the parts out alternately -- transforms, updates, colour -- measured slower: 9.5 -> 10.4 us;
  // about 770 of these 8-wavefront workgroups are resident at a time, so what matters is how MANY there are: the second mapper brings few, see the host side)
  const int32_t g = (int32_t)blockIdx.x;
  const bool is_a = g < a.n_wg; const int32_t wg = is_a ? g : g - a.n_wg;
  if (is_a)
    integrate_tsdf_color_body<Img, Pix, 1, false>(a.m, a.fs, CameraSensor{}, a.view_list, a.list_cap, a.mesh_list, a.view_export, a.view_export_cap, a.spec_lanes, a.n_tsdf_wg, a.fsc, a.synth, a.srows,
                                                  a.scols, a.cand, a.cand_cnt_idx, a.n_edt_wg, a.ea, a.imp, wg, a.n_wg, smem);
  else
    integrate_tsdf_color_body<Img, Pix, 1, false>(b.m, b.fs, CameraSensor{}, b.view_list, b.list_cap, b.mesh_list, b.view_export, b.view_export_cap, b.spec_lanes, b.n_tsdf_wg, b.fsc, b.synth, b.srows,
                                                  b.scols, b.cand, b.cand_cnt_idx, b.n_edt_wg, b.ea, b.imp, wg, b.n_wg, smem);
}
static_assert(2 * sizeof(FusedArgs<DepthF32, PixRgb8>) <= 4096, "k_integrate_tsdf_color_pair: kernel arguments");
// (a depth batch AND a colour batch in one argument block: the 4 KiB kernel-argument limit is why the colour path's frames are FrameCore)
static_assert(sizeof(DMap) + sizeof(FrameSet<DepthF32, MAX_BATCH>) + sizeof(FrameSetC<PixRgb8, MAX_BATCH>) + sizeof(EsdfArgs) + sizeof(ImportArgs) + 160 <= 4096, "k_integrate_tsdf_color<.., MAX_BATCH>: kernel arguments");
static_assert(sizeof(DMap) + sizeof(FrameSet<DepthF32, MAX_BATCH>) + sizeof(TraceRiderT<MAX_BATCH>) + sizeof(EsdfArgs) + 64 <= 4096, "k_mark_view<.., MAX_BATCH>: kernel arguments");

int nvbx_mapper::ensure_fuse_buffers() {
  if (fuse_cap == capacity && color_cand) return NVBX_OK;
  NVBX_HIP(hipStreamSynchronize(stream));
  if (color_cand) NVBX_HIP(hipFree(color_cand));
  color_cand = nullptr; fuse_cap = 0;
  NVBX_HIP(hipMalloc(&color_cand, (size_t)capacity * 2 * sizeof(int4)));
  fuse_cap = capacity;
  return NVBX_OK;
}

// ------------------------------------------------------------------------------------------------ LiDAR, far field: beam-centric update
// Measured on a configs[4] scan (an instrumented copy of the CPU checker): 78 % of the voxels of the blocks in view run the nearest-beam rule
// -- the four beams around them do not agree, as on a ground plane seen at a grazing angle -- and only 6 % pass it: at 0.10 m voxels a beam's
// acceptance tube is one voxel wide while the beams are 0.6 m x 1.2 m apart at 100 m.  One lane per voxel pays the projection, four taps, the
// nearest tap and the point-to-ray distance 512 times per block to update ~20 voxels.  This launch turns the question round for the blocks
// where ONLY the nearest-beam rule can apply: ONE WAVEFRONT PER BLOCK
//   (1) projects the block's 8 corners: its footprint in the range image (+ 0.75 px: the elevation of a box is not extremal at its corners);
//       a block whose corner fails to project, that straddles the azimuth seam, or whose footprint exceeds 64 pixels is left to the dense launch;
//   (2) tests every 2 x 2 beam quad a voxel of the block could interpolate in ("four returns that agree"): one valid quad -> dense launch;
//   (3) otherwise walks every beam of the footprint that has a return through the block: in block voxel coordinates the beam is a line, along its
//       major axis it crosses 8 voxel slices, and a voxel centre within (0.5 + 0.01) voxel of the line lies within 0.51 / 0.577 = 0.88 < 1 voxel of
//       the crossing point inside its slice, i.e. among the 2 x 2 cells around that point -- 32 candidate voxels per beam instead of 512 per block;
//   (4) evaluates every candidate with the SAME per-voxel code as the dense launch (LidarSensor::sample_px, tsdf_fuse_plain; the voxel centre is
//       block origin + rotated offset, exactly as there) and updates it iff the rule's nearest beam is the beam that enumerated it (so a voxel
//       near two tubes is updated once).  A voxel that is not a candidate of its nearest beam fails that beam's distance test: untouched, as in the
//       dense launch.  Result: bit-identical maps (tests/test_gpu_full_size.py compares all 112 k blocks of two scans with the CPU checker).
// The class of every record (1 = updated here) goes to view_class[]; the dense launch that follows skips those.  Requires the plain integrator
// configuration (constant weighting, TSDF) and a nearest-beam acceptance radius <= 0.55 voxel (the 2 x 2 argument); the host falls back otherwise.
#ifndef NVBX_SPARSE_WAVES
#define NVBX_SPARSE_WAVES 6      // (81 VGPRs were one register away from six wavefronts per SIMD: 5 / 6 / 7 / 8 asked for = 116.6 / 110.0 / 111.1 / 114.4 us)
#endif
template <typename Img>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NVBX_SPARSE_WAVES, NVBX_SPARSE_WAVES))) void k_lidar_sparse(DMap m, FrameSet<Img, 1> fs, LidarSensor sensor, const int4* view_list, int32_t list_cap,
                                                      int32_t mesh_list, uint8_t* view_class, int32_t* dense_list) {
  // per wavefront: the crossing beams {line in block voxel coordinates ob[3], db[3]; pixel; range; direction[3]; major axis} and the
  // candidate voxels that survive the geometric pre-filter {beam << 9 | voxel}
  __shared__ float s_beam[4][64][12];
  __shared__ uint16_t s_item[4][512];
  const Frame& f = fs.f[0];
  const Img& img = fs.img[0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t n_waves = (int32_t)gridDim.x * 4;
  int32_t n = m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  if (n > list_cap) n = list_cap;
  const float vs = f.voxel_size, bs = f.block_size;
  const int rows = f.rows, cols = f.cols;
  // Dependent-access chain per block: {record (fetched one block ahead)} -> {quad taps || footprint taps || beam tables || flag atomic} ->
  // {voxels of the candidates that pass} -> store.  The candidates' arithmetic needs no image access at all: their beam's range and direction
  // travel with the beam.
  // Records are taken EIGHT at a time: lanes 8 j .. 8 j + 7 project the eight corners of block j, so the footprints of eight blocks cost one
  // pass of the projection code (with one block per pass 56 of the 64 lanes idled through it); the blocks are then walked one after the other.
  const int grp = lane >> 3;
  int32_t n_mine = 0;                                        // blocks this wavefront updated (one counter atomic per wavefront, at the end)
  // pass p of G = ceil(n / 8) takes records 8 p .. 8 p + 7 -- or, NVBX_LIDAR_SPARSE_STRIDED, records p, G + p, 2 G + p, ...: eight far-apart
  // places of the list (a pass's cost is the sum of its blocks' footprints; neighbours in the list have footprints of one size)
  const int32_t G = (n + 7) >> 3;
  auto ridx = [&](int32_t p, int g) -> int32_t { return NVBX_LIDAR_SPARSE_STRIDED ? g * G + p : p * 8 + g; };
  int32_t pass = (int32_t)blockIdx.x * 4 + wv;
  int4 rec_next = (pass < G && ridx(pass, grp) < n) ? view_list[ridx(pass, grp)] : make_int4((int32_t)SLOT_NONE, 0, 0, 0);
  for (; pass < G; pass += n_waves) {
    const int4 rec_g = rec_next;
    if (pass + n_waves < G && ridx(pass + n_waves, grp) < n) rec_next = view_list[ridx(pass + n_waves, grp)]; else rec_next = make_int4((int32_t)SLOT_NONE, 0, 0, 0);
    // (1) corners of the group's block
    bool sparse_g = ridx(pass, grp) < n && slot_ok((uint32_t)rec_g.x);
    float org_g[3];
    sensor_block_origin(f, rec_g.y, rec_g.z, rec_g.w, org_g);
    int c0_g = 0, r0_g = 0, w_g = 0, h_g = 0;
    {
      float off[3], pc[3], u = 0.0f, v = 0.0f;
      rotate(f.R_CL, (float)(lane & 1) * bs, (float)((lane >> 1) & 1) * bs, (float)((lane >> 2) & 1) * bs, off);
      pc[0] = org_g[0] + off[0]; pc[1] = org_g[1] + off[1]; pc[2] = org_g[2] + off[2];
      const bool okc = nvbx_lidar_project(&sensor.l, pc, nvbx_lidar_range(pc), &u, &v) != 0;
      const u64 bad = __ballot(!okc);
      if ((bad >> (8 * grp)) & 0xFFull) sparse_g = false;
      float umin = u, umax = u, vmin = v, vmax = v;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        umin = fminf(umin, __shfl_xor(umin, o)); umax = fmaxf(umax, __shfl_xor(umax, o));
        vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o));
      }
      if (sparse_g) {
        if (umax - umin > (float)cols * 0.5f) sparse_g = false;               // straddles the azimuth seam
        c0_g = (int)floorf(umin - 0.75f); r0_g = (int)floorf(vmin - 0.75f);
        w_g = (int)floorf(umax + 0.75f) - c0_g + 1; h_g = (int)floorf(vmax + 0.75f) - r0_g + 1;
        if (w_g < 1 || h_g < 1 || w_g * h_g > 64) sparse_g = false;           // one lane per footprint pixel
      }
    }
    const u64 sparse_groups = __ballot(sparse_g);
    uint32_t dmask = 0;                                                         // records of this pass left to the dense launch (uniform)
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
    const int32_t i = ridx(pass, j);
    if (i >= n) continue;                                                       // (uniform)
    bool sparse = ((sparse_groups >> (8 * j)) & 1ull) != 0;
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane(rec_g.x, 8 * j);
    if (!sparse) { if (lane == 0) view_class[i] = 0; if (slot_ok(slot)) dmask |= 1u << j; continue; }               // (uniform)
    const float org[3] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(org_g[0]), 8 * j)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org_g[1]), 8 * j)),
                          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org_g[2]), 8 * j))};
    const int c0 = __builtin_amdgcn_readlane(c0_g, 8 * j), r0 = __builtin_amdgcn_readlane(r0_g, 8 * j), w = __builtin_amdgcn_readlane(w_g, 8 * j), h = __builtin_amdgcn_readlane(h_g, 8 * j);
    // footprint pixel of this lane: its range and its beam's direction tables are requested together with the quad taps below
    // (lane -> (column, row) of a w-wide grid without an integer division: exact for lane < 2^10)
    const int ly = (int)(((float)lane + 0.5f) * (1.0f / (float)w)), lx = lane - ly * w;
    const int bc = c0 + lx, brr = r0 + ly;
    const bool in_img = ly < h && bc >= 0 && brr >= 0 && bc < cols && brr < rows;
    const float bd = in_img ? img(pix(brr, bc, cols)) : 0.0f;
    const float2 te = sensor.el_tab[in_img ? brr : 0], ta = sensor.az_tab[in_img ? bc : 0];
    // (2) quads x0 in [c0 - 1, c0 + w - 1], y0 in [r0 - 1, r0 + h - 1]: (w + 1) x (h + 1) <= 130 of them, up to three per lane
    bool anyq = false;
    const int nq = (w + 1) * (h + 1);
    const float iw1 = 1.0f / (float)(w + 1);
    for (int qi = lane; qi < nq; qi += 64) {
      const int qy = (int)(((float)qi + 0.5f) * iw1), qx = qi - qy * (w + 1);
      const int x0 = c0 - 1 + qx, y0 = r0 - 1 + qy;
      if (!(x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1)) {
        const int32_t i00 = pix(y0, x0, cols);
        const float f00 = img(i00), f10 = img(i00 + 1), f01 = img(i00 + cols), f11 = img(i00 + cols + 1);
        if (f00 > 0.0f && f10 > 0.0f && f01 > 0.0f && f11 > 0.0f) {
          const float mx = fmaxf(fmaxf(f00, f10), fmaxf(f01, f11)), mn = fminf(fminf(f00, f10), fminf(f01, f11));
          if (mx - mn <= sensor.max_diff_m) anyq = true;
        }
      }
    }
    if (__ballot(anyq)) sparse = false;
    if (!sparse) { if (lane == 0) view_class[i] = 0; dmask |= 1u << j; continue; }               // (uniform)
    // (3) the beams of the footprint that have a return, one per lane: line in block voxel coordinates q = R_LC (P - org) / vs (sensor origin: P = 0)
    float ob[3] = {0.0f, 0.0f, 0.0f}, db[3] = {1.0f, 0.0f, 0.0f};
    const float dir[3] = {te.y * ta.y, te.y * ta.x, te.x};                   // == LidarSensor::beam_dir(brr, bc)
    bool crossing = false; int axis = 0;
    if (in_img && bd > 0.0f) {
      float t3[3];
      rotate(f.R_LC, org[0], org[1], org[2], t3);
      const float ivs = 1.0f / vs;
      ob[0] = -t3[0] * ivs; ob[1] = -t3[1] * ivs; ob[2] = -t3[2] * ivs;    // (enumeration geometry only: a slack of 0.1 voxel, no exactness needed)
      rotate(f.R_LC, dir[0], dir[1], dir[2], db);
      int a = 0; if (fabsf(db[1]) > fabsf(db[a])) a = 1; if (fabsf(db[2]) > fabsf(a == 0 ? db[0] : db[1])) a = 2;
      axis = a;
      const float oa = a == 0 ? ob[0] : (a == 1 ? ob[1] : ob[2]), da = a == 0 ? db[0] : (a == 1 ? db[1] : db[2]);
      const float obb = a == 0 ? ob[1] : (a == 1 ? ob[2] : ob[0]), dbb = a == 0 ? db[1] : (a == 1 ? db[2] : db[0]);     // axes (a + 1) % 3, (a + 2) % 3
      const float occ = a == 0 ? ob[2] : (a == 1 ? ob[0] : ob[1]), dcc = a == 0 ? db[2] : (a == 1 ? db[0] : db[1]);
      const float ida = 1.0f / da;
      // the line's crossing points of slice 0 and slice 7 bound those of the slices between: the tube meets the block iff the interval of
      // crossing points (+- one cell) meets [0, 7] in both perpendicular axes
      const float t0 = (0.5f - oa) * ida, t7 = (7.5f - oa) * ida;
      const float b0 = obb + t0 * dbb - 0.5f, b7 = obb + t7 * dbb - 0.5f, cc0 = occ + t0 * dcc - 0.5f, cc7 = occ + t7 * dcc - 0.5f;
      crossing = fmaxf(b0, b7) >= -1.0f && fminf(b0, b7) <= 8.0f && fmaxf(cc0, cc7) >= -1.0f && fminf(cc0, cc7) <= 8.0f;
    }
    const u64 cross = __ballot(crossing);
    const int nb = (int)__popcll(cross);
    if (crossing) {
      const int k = (int)__popcll(cross & ((1ull << lane) - 1ull));
      float* q = s_beam[wv][k];
      q[0] = ob[0]; q[1] = ob[1]; q[2] = ob[2]; q[3] = db[0]; q[4] = db[1]; q[5] = db[2];
      q[6] = __int_as_float(pix(brr, bc, cols)); q[7] = bd; q[8] = dir[0]; q[9] = dir[1]; q[10] = dir[2]; q[11] = __int_as_float(axis);
    }
    __threadfence_block();                                  // (the wavefront's own LDS writes, read by its other lanes below: no workgroup barrier -- the four wavefronts run independent loops)
    __builtin_amdgcn_wave_barrier();
    // (4a) candidates: item = (beam, slice, cell) -> 32 per beam; a candidate survives the PRE-FILTER if its centre lies within the acceptance radius + 0.02 voxel of the
    // beam's line in block coordinates (the exact test below is this distance in the sensor frame, to ~1e-4 voxel); survivors are compacted
    int32_t ns = 0;                                           // survivors (wave-uniform)
    const float pre = sensor.max_ray_dist_m / vs + 0.02f, pre2 = pre * pre;     // acceptance radius of the exact test, in voxels, + slack
    for (int it0 = 0; it0 < nb * 32; it0 += 64) {
      const int item = it0 + lane;
      const int kb = item >> 5, sl = (item >> 2) & 7, cell = item & 3;
      bool keep = false; int vox = 0;
      if (kb < nb) {
        const float* q = s_beam[wv][kb];
        const float o0 = q[0], o1 = q[1], o2 = q[2], d0 = q[3], d1 = q[4], d2 = q[5];
        const int a = __float_as_int(q[11]);
        const float oa = a == 0 ? o0 : (a == 1 ? o1 : o2), da = a == 0 ? d0 : (a == 1 ? d1 : d2);
        const float obb = a == 0 ? o1 : (a == 1 ? o2 : o0), dbb = a == 0 ? d1 : (a == 1 ? d2 : d0);
        const float occ = a == 0 ? o2 : (a == 1 ? o0 : o1), dcc = a == 0 ? d2 : (a == 1 ? d0 : d1);
        const float t = ((float)sl + 0.5f - oa) / da;
        const int jb = (int)floorf(obb + t * dbb - 0.5f) + (cell & 1), jc = (int)floorf(occ + t * dcc - 0.5f) + (cell >> 1);
        if (jb >= 0 && jb <= 7 && jc >= 0 && jc <= 7) {
          const int vx = a == 0 ? sl : (a == 1 ? jc : jb), vy = a == 0 ? jb : (a == 1 ? sl : jc), vz = a == 0 ? jc : (a == 1 ? jb : sl);
          const float px = (float)vx + 0.5f - o0, py = (float)vy + 0.5f - o1, pz = (float)vz + 0.5f - o2;
          const float cx = py * d2 - pz * d1, cy = pz * d0 - px * d2, cz = px * d1 - py * d0;     // |(p - o) x d|^2 = squared distance to the line (|d| = 1)
          keep = (cx * cx + cy * cy) + cz * cz <= pre2;
          vox = vz + 8 * vy + 64 * vx;
        }
      }
      const u64 km = __ballot(keep);
      if (keep) { const int pos = ns + (int)__popcll(km & ((1ull << lane) - 1ull)); if (pos < 512) s_item[wv][pos] = (uint16_t)((kb << 9) | vox); }
      ns += (int32_t)__popcll(km);
    }
    if (ns > 512) sparse = false;                             // more survivors than the list holds (dense beams at close range): the dense launch takes the block
    if (lane == 0) view_class[i] = sparse ? 1 : 0;
    if (!sparse) { dmask |= 1u << j; __builtin_amdgcn_wave_barrier(); continue; }                // (uniform)
    // the block's books, as the dense launch keeps them (lane 0; the returning atomic is consumed after the update)
    uint32_t old = 0;
    if (lane == 0) old = atomicOr(&m.slot_flags[slot], F_TSDF | F_DIRTY_ESDF | F_DIRTY_MESH | F_BAND_STALE);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // (4b) the survivors, 64 at a time.  In a block of this class the four-tap rule cannot measure (no valid quad in reach), so a candidate goes
    // straight to the nearest-beam rule -- the same operations, in the same order, as the tail of LidarSensor::sample_px, with the beam's range and
    // direction taken from the beam instead of from the image and the tables.
    for (int it0 = 0; it0 < ns; it0 += 64) {
      if (it0 + lane >= ns) continue;
      const int code = s_item[wv][it0 + lane];
      const float* q = s_beam[wv][code >> 9];
      const int vox = code & 511, vx = vox >> 6, vy = (vox >> 3) & 7, vz = vox & 7;
      float off[3], pc[3];
      sensor_voxel_offset(f, vx, vy, vz, off);
      pc[0] = org[0] + off[0]; pc[1] = org[1] + off[1]; pc[2] = org[2] + off[2];
      const float r = nvbx_lidar_range(pc);
      if (f.max_dist > 0.0f && r > f.max_dist) continue;
      float u, v;
      if (!nvbx_lidar_project(&sensor.l, pc, r, &u, &v)) continue;
      const int c = (int)floorf(u), rr = (int)floorf(v);
      if (c < 0 || rr < 0 || c >= cols || rr >= rows) continue;
      if (pix(rr, c, cols) != __float_as_int(q[6])) continue;   // the rule's nearest beam is another one: that beam's walk takes the voxel (if it can)
      const float d = q[7];
      const float bdx = q[8], bdy = q[9], bdz = q[10];
      const float dot = __builtin_fmaf(pc[2], bdz, __builtin_fmaf(pc[1], bdy, pc[0] * bdx));
      const float ex = __builtin_fmaf(-dot, bdx, pc[0]), ey = __builtin_fmaf(-dot, bdy, pc[1]), ez = __builtin_fmaf(-dot, bdz, pc[2]);
      if (__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) > sensor.max_ray_dist_m * sensor.max_ray_dist_m) continue;
      float2* vp = &m.tsdf[(size_t)slot * 512 + vox];
      float2 fin = *vp;
      if (tsdf_fuse_plain(f, &fin, d, r)) *vp = fin;
    }
    if (lane == 0) {
      if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)slot);
      if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, (int32_t)slot);
    }
    n_mine++;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    }
    // the dense launch's work list: the view-list indices of the records left to it, one reservation per pass in this workgroup's shard
    if (dense_list && dmask) {
      const int sh = my_shard();
      int32_t base0 = 0;
      if (lane == 0) base0 = atomicAdd(shc_at(m, S_LIDAR_SPARSE, sh, 1), (int32_t)__popc(dmask));
      base0 = __shfl(base0, 0);
      if (lane < 8 && ((dmask >> lane) & 1u)) {
        const int32_t pos = base0 + (int32_t)__popc(dmask & ((1u << lane) - 1u));
        if (pos < list_cap) dense_list[(size_t)sh * list_cap + pos] = ridx(pass, lane);
      }
    }
  }
  if (lane == 0 && n_mine) atomicAdd(shc_at(m, S_LIDAR_SPARSE, my_shard(), 0), n_mine);
}

// the beam-centric far-field launch (LiDAR only); view_class = nullptr: everything goes to the dense launch
template <typename Img, typename Sensor, int NB>
static int launch_lidar_sparse(nvbx_mapper*, const FrameSet<Img, NB>&, const Sensor&, bool, uint8_t** view_class, int32_t** dense_list) { *view_class = nullptr; *dense_list = nullptr; return NVBX_OK; }
template <typename Img>
static int launch_lidar_sparse(nvbx_mapper* m, const FrameSet<Img, 1>& fs, const LidarSensor& sensor, bool plain, uint8_t** view_class, int32_t** dense_list) {
  *view_class = nullptr; *dense_list = nullptr;
  static const int enabled = getenv("NVBX_LIDAR_SPARSE") ? atoi(getenv("NVBX_LIDAR_SPARSE")) : 1;       // (A/B: 0 = dense launch only)
  if (!enabled || !plain || !(m->p.lidar_nearest_interpolation_max_allowable_dist_to_ray_vox <= 0.55f)) return NVBX_OK;
  if (m->view_class_cap < m->capacity) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->view_class) NVBX_HIP(hipFree(m->view_class));
    m->view_class = nullptr; m->view_class_cap = 0;
    // [capacity class bytes][NSH x capacity view-list indices: the dense launch's work list, one region per shard]
    NVBX_HIP(hipMalloc(&m->view_class, (((size_t)m->capacity + 15) & ~(size_t)15) + (size_t)NSH * (size_t)m->capacity * 4));
    m->view_class_cap = m->capacity;
  }
  static const int sparse_grid = getenv("NVBX_LIDAR_SPARSE_GRID") ? atoi(getenv("NVBX_LIDAR_SPARSE_GRID")) : 2048;    // (six resident wavefronts per SIMD = 1536 workgroups; 1536 / 2048 / 2560 / 3072 / 3584 / 4096 / 8192 workgroups: 111.1 / 109.5 / 110.5 / 111.0 / 114.8 / 115.3 / 114.3 us with strided passes and the work list)
  // (with an exchange buffer registered -- nvbx_set_view_export -- the dense launch walks the whole view list, as it writes every record's index there)
  static const int use_list = getenv("NVBX_LIDAR_DENSE_LIST") ? atoi(getenv("NVBX_LIDAR_DENSE_LIST")) : 1;       // (A/B: 0 = the dense launch skips the taken records of the whole list)
  int32_t* dense = (use_list && !m->view_export) ? reinterpret_cast<int32_t*>(m->view_class + (((size_t)m->capacity + 15) & ~(size_t)15)) : nullptr;
  NVBX_LAUNCH(m, (k_lidar_sparse<Img>), dim3(sparse_grid), dim3(256), m->d, fs, sensor, (const int4*)m->view_list, (int32_t)m->capacity, m->mesh_list_live(), m->view_class, dense);
  *dense_list = dense;
  *view_class = m->view_class;
  return NVBX_OK;
}

// LiDAR view calculation over the dense grid (k_mark_view_grid, k_scan_view_grid, k_resolve_view) instead of k_mark_view; *used = false: the caller
// launches k_mark_view (cameras; a scan without a range limit or with a box beyond the grid's addressing / memory cap; NVBX_LIDAR_VIEW_GRID=0)
template <typename Img, typename Sensor, int NB>
static int launch_view_grid(nvbx_mapper*, const FrameSet<Img, NB>&, const Sensor&, int, int32_t, bool* used) { *used = false; return NVBX_OK; }
template <typename Img>
static int launch_view_grid(nvbx_mapper* m, const FrameSet<Img, 1>& fs, const LidarSensor& sensor, int tiles, int32_t fence_report, bool* used) {
  *used = false;
  static const int enabled = getenv("NVBX_LIDAR_VIEW_GRID") ? atoi(getenv("NVBX_LIDAR_VIEW_GRID")) : 1;       // (A/B: 0 = k_mark_view<Lidar>)
  const Frame& f = fs.f[0];
  if (!enabled || !(f.max_dist > 0.0f) || m->capacity > (1ll << 24)) return NVBX_OK;
  // the box: every ray ends within max_dist of the sensor; along z the beams' elevation range bounds it (|world z of a unit beam| <=
  // hypot(R20, R21) cos(el) + |R22 sin(el)|, elevation table rows on the host: ensure_lidar_tables)
  const double reach = (double)f.max_dist / (double)f.block_size;
  double wz = 0.0;
  const double hxy = std::hypot((double)f.R_LC[6], (double)f.R_LC[7]);
  for (int k = 0; k < sensor.l.rows; k++) wz = std::max(wz, hxy * std::fabs((double)m->lidar_host[2 * (size_t)k + 1]) + std::fabs((double)f.R_LC[8] * (double)m->lidar_host[2 * (size_t)k]));
  int64_t H = (int64_t)std::ceil(reach) + 2, Hz = std::min<int64_t>(H, (int64_t)std::ceil(reach * std::min(1.0, wz)) + 2);
  // (tests: a box SMALLER than the sensor's range -- the blocks beyond it take the hash path, block by block; tests/test_gpu_round5.py)
  static const int64_t reach_cap = getenv("NVBX_VIEW_GRID_REACH") ? atoll(getenv("NVBX_VIEW_GRID_REACH")) : 0;
  if (reach_cap > 0) { H = std::min(H, reach_cap); Hz = std::min(Hz, reach_cap); }
  const int64_t ncx = (2 * H + 1 + 3) / 4, ncz = (2 * Hz + 1 + 3) / 4;
  const int64_t cells = ncx * ncx * ncz;
  static const int64_t cap_mb = getenv("NVBX_VIEW_GRID_MAX_MB") ? atoll(getenv("NVBX_VIEW_GRID_MAX_MB")) : 128;
  if (ncx > 256 || ncz > 256 || cells * 64 > (cap_mb << 20)) return NVBX_OK;
  const size_t coarse_bytes = ((size_t)cells + 3) & ~(size_t)3;
  if (m->view_grid_cells_cap < cells) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->view_grid_fine) NVBX_HIP(hipFree(m->view_grid_fine));
    m->view_grid_fine = nullptr; m->view_grid_cells_cap = 0;
    NVBX_HIP(hipMalloc(&m->view_grid_fine, (size_t)cells * 64 + coarse_bytes));      // [fine: 64 B per cell][coarse: 1 B per cell]
    m->view_grid_cells_cap = cells; m->view_grid_dirty = true;
  }
  if (m->view_grid_dirty) NVBX_HIP(hipMemsetAsync(m->view_grid_fine, 0, (size_t)m->view_grid_cells_cap * 64 + (((size_t)m->view_grid_cells_cap + 3) & ~(size_t)3), m->stream));
  m->view_grid_dirty = true;                 // until all three launches are enqueued
  ViewGrid vg{};
  vg.fine = m->view_grid_fine; vg.coarse = m->view_grid_fine + (size_t)m->view_grid_cells_cap * 64;
  vg.cx = (int32_t)std::floor(f.t_LC[0] / f.block_size); vg.cy = (int32_t)std::floor(f.t_LC[1] / f.block_size); vg.cz = (int32_t)std::floor(f.t_LC[2] / f.block_size);
  vg.ox = vg.cx - (int32_t)H; vg.oy = vg.cy - (int32_t)H; vg.oz = vg.cz - (int32_t)Hz;
  vg.ncx = (int32_t)ncx; vg.ncy = (int32_t)ncx; vg.ncz = (int32_t)ncz;
  vg.tag = 0x80000000u | f.frame_id;
  NVBX_LAUNCH(m, (k_mark_view_grid<Img>), dim3(tiles), dim3(64), m->d, fs, sensor, (int4*)m->view_list, (int32_t)m->capacity, (int32_t)(m->premark_consumed ? 1 : 0), fence_report, vg);
  // the resolving launch: one tagged record per lane -- as many as the last finished scan had in view (+ 25 %; a hint only, it grid-strides)
  // the scan: a wavefront per four lines (256 cells) of the coarse map, VG_SCAN_WAVES wavefronts per workgroup, everything in one pass
  const int64_t coarse_lines = ((cells + 3) / 4 + 15) / 16;
  const int64_t scan_wg = (coarse_lines + 4 * VG_SCAN_WAVES - 1) / (4 * VG_SCAN_WAVES);
  NVBX_LAUNCH(m, k_scan_view_grid, dim3((unsigned)scan_wg), dim3(64 * VG_SCAN_WAVES), m->d, f.frame_id, (int4*)m->view_list, (int32_t)m->capacity, vg);
  const int64_t n_hint = std::max<int64_t>(0, __atomic_load_n(&m->h_mirror[2], __ATOMIC_RELAXED));
  const int64_t rec_wg = n_hint == 0 ? 512 : std::max<int64_t>(8, std::min<int64_t>(2048, (n_hint + n_hint / 4 + 255) / 256));
  NVBX_LAUNCH(m, k_resolve_view, dim3((unsigned)rec_wg), dim3(256), m->d, f.frame_id, (int4*)m->view_list, (int32_t)m->capacity, vg.tag);
  NVBX_HIP(hipGetLastError());
  m->view_grid_dirty = false;
  *used = true;
  return NVBX_OK;
}

// ---- One depth frame's two launches, in STEPS (round 6): integrate_depth_impl runs them in order for one mapper; nvbx_integrate_depth_pair interleaves the
// steps of TWO mappers around two shared launches (k_mark_view_pair, k_integrate_tsdf_color_pair).  The steps are the former body of integrate_depth_impl, cut
// where it launches; what each step does to the mapper's host state, and in which order, is unchanged.
static bool fused_colour_applies(const nvbx_mapper* m) {
  return m->p.projective_layer_type != 1 && m->p.esdf_mode == 0 && m->p.esdf_propagation == 0 && !m->lidar_integrated && m->capacity <= (1ll << 24);
}
template <int NB> struct DepthSteps {
  int tiles = 0, edt_wg = 0; EsdfArgs ea{}; TraceRiderT<NB> tr{};                 // launch 1
  bool plain = true, has_color = false, pipelined = false, fused = false;
  FrameSetC<PixRgb8, NB> fsc{}; int f_kind = 0; int32_t f_srows = 0, f_scols = 0;      // launch 2 (fused form)
  int grid = 8; int32_t spec_lanes = 1;
  int32_t n_edt = 0; EsdfArgs ea_edt{}; const int4* cand = nullptr; int32_t cand_idx = 0; int cgrid = 0; ImportArgs imp{};
};
// step 1: everything in front of the view-marking launch (ray grid, riders of the held-back calls, the fence report)
template <typename Img, typename Sensor, int NB>
static int depth_step_before_mark_view(nvbx_mapper* m, FrameSet<Img, NB>& fs, DepthSteps<NB>& st) {
  const int s = fs.f[0].subsample;
  for (int c = 0; c < fs.n; c++) {
    fs.f[c].n_ray_rows = (fs.f[c].rows + s - 1 + s - 1) / s;   // indices i with i*s < rows + s - 1
    fs.f[c].n_ray_cols = (fs.f[c].cols + s - 1 + s - 1) / s;
    fs.f[c].cam_bit = 1u << c;
  }
  const Frame& f = fs.f[0];
  st.tiles = mark_view_tile_wgs<Sensor>(f) * fs.n;       // tile workgroups (padded: the groups of one XCD are a contiguous band, k_mark_view); camera after camera
  // a held-back EDT rides in this launch (camera: 256-thread workgroups); the LiDAR launch is 64 threads wide, so flush first
  st.edt_wg = 0; st.ea = m->edt_args;
  if (m->edt_pending) {
    if (Sensor::kRiders) { st.edt_wg = 256; m->edt_pending = false; }        // (256 .. 1024 riders measured: no difference, profiles/r02x_kernel_isolation.txt)
    else if (m->flush_edt()) return NVBX_E_DEVICE;
  }
  // Colour deferral: a held-back integrateColor (and an updateEsdf behind it) is carried out in PIPELINED order -- its sphere tracing rides
  // in this view-marking launch, its colour integration + ESDF marking follow, then this frame's TSDF update: three launches per frame.
  // Or TWO (fused, k_integrate_tsdf_color above): the colour frame's candidate blocks are discovered and the ESDF marking pass runs as
  // riders of this view-marking launch too, and colour integration, the update's distance transform and this frame's TSDF update share
  // the second launch.
  // A held-back updateEsdf with NO colour frame in front (depth-only hosts, occupancy mappers) is carried the same way: marking pass here,
  // distance transform in the second launch, no colour workgroups.
  st.tr = TraceRiderT<NB>{};
  st.plain = true;
  for (int c = 0; c < fs.n; c++) st.plain = st.plain && frame_is_plain(fs.f[c]);
  st.has_color = m->color_pending.on;
  // (one frame carries a frame, a batch a batch; a held-back updateEsdf WITHOUT a colour frame -- depth-only and occupancy mappers -- is carried by
  //  any camera launch: integrate_cameras has checked that the two-launch order applies, nvbx_mapper::esdf_only_carry)
  st.pipelined = Sensor::kRiders && (st.has_color ? ((NB == 1) == (m->color_pending.n == 1)) : m->esdf_update_pending);
  st.fused = false;
  if (st.pipelined) {
    static const int fuse_on = getenv("NVBX_FUSE_COLC") ? atoi(getenv("NVBX_FUSE_COLC")) : 1;      // (A/B: 0 = three launches per frame)
    // TSDF mapper (with or without a freespace layer), 2-D ESDF by the exact transform (the marking pass / distance transform that ride are the
    // 2-D ones), no multi-GPU union step waiting for the colour launch, and no block that may be F_BAND_STALE (the candidate riders read the
    // band flags only)
    st.fused = st.has_color ? (fuse_on && fused_colour_applies(m))
                            : true;      // (no colour: no candidates, no band flags -- esdf_only_carry has checked the rest)
    // (a distance transform armed outside the pipeline must precede the marking pass that rides in this launch: its own launch, rare)
    if (st.fused && st.edt_wg) { m->edt_pending = true; st.edt_wg = 0; if (m->flush_edt()) return NVBX_E_DEVICE; }
    m->pipelined_order = true;
    if (st.has_color) { const int rc = m->pending_color_trace_rider(&st.tr); if (rc) { m->pipelined_order = false; return rc; } }
    // riders before or after the tiles (A/B: NVBX_MARK_TILES_FIRST = 0 / 1).  One frame: tiles first (15.2 vs 15.8 us).  A batch of 8: riders first
    // (32.4 vs 42.0 us) -- its 2 688 single-wavefront tile workgroups, each holding its LDS key set, take most of the workgroup slots, and
    // sphere-tracing workgroups dispatched behind them start when the tiles are done: the launch took the SUM of its parts.
    static const int tiles_first_env = getenv("NVBX_MARK_TILES_FIRST") ? atoi(getenv("NVBX_MARK_TILES_FIRST")) : -1;
    const bool tiles_first = tiles_first_env >= 0 ? tiles_first_env != 0 : NB == 1;
    if (tiles_first) st.tr.n_tile_wg = st.tiles;
    if (st.fused && !st.has_color) m->pending_marking_args(&st.tr.n_mark_wg, &st.ea, NB == 1);
    if (st.fused && st.has_color) {
      if (m->ensure_fuse_buffers()) { m->pipelined_order = false; return NVBX_E_DEVICE; }
      const int64_t hw_seen = std::max<int64_t>(1, __atomic_load_n(&m->h_mirror[1], __ATOMIC_RELAXED));
      st.tr.n_scan_wg = (int32_t)std::min<int64_t>(256, 8 * ((hw_seen + hw_seen / 4 + 64 + 2047) / 2048));      // 256 slots per workgroup and pass; a hint only (the riders grid-stride)
      st.tr.cand = m->color_cand + (size_t)m->cand_parity * m->fuse_cap;
      st.tr.cand_cnt_idx = C_CAND_COUNT + m->cand_parity; st.tr.cand_reset_idx = C_CAND_COUNT + (1 - m->cand_parity);
      m->cand_parity ^= 1;      // (the next fused launch resets THIS count, whether or not the colour launch below is reached: an error return in between leaves no stale candidates behind)
      m->pending_marking_args(&st.tr.n_mark_wg, &st.ea, NB == 1);        // (the held-back integrateColor's marking pass, in call order: before its colour integration below)
    }
  }
  st.tr.fence_report = m->next_fence_report();
  return NVBX_OK;
}
// step 2: between the two launches -- the host-side steps of the held-back calls, then the sizes of the TSDF-update part
template <typename Sensor, int NB>
static int depth_step_between(nvbx_mapper* m, DepthSteps<NB>& st) {
  if (st.pipelined) {
    // the host-side steps of the held-back calls, in call order: integrateColor (its marking pass empties the dirty list itself, the EDT
    // of the update keeps it -- EsdfArgs), then updateEsdf (which only arms the next held-back EDT: the marking pass has been launched)
    if (!st.fused) m->premark_consumed = false;
    int rc = NVBX_OK;
    if (st.fused) { if (st.has_color) rc = m->pending_color_fused_args(&st.fsc, &st.f_kind, &st.f_srows, &st.f_scols); }
    else rc = m->launch_pending_color_after_trace();
    if (rc == NVBX_OK && m->esdf_update_pending) { m->esdf_update_pending = false; rc = nvbx_update_esdf(m); }
    m->pipelined_order = false;
    if (rc) return rc;
  }
  m->premark_consumed = false; m->dirty_since_mark = true;
  // grid-stride over the view list: exactly the 1024 workgroups that are resident together (4 per CU)
  // (a camera BATCH: 512 -- its fused launch is residency-bound, 1 024 eight-wavefront workgroups resident, and 1 024 TSDF workgroups in front kept the colour
  //  part waiting: 4 / 8 cameras 0.0382 / 0.0560 -> 0.0365 / 0.0547 ms per step, tools/fused_grid_sweep.sh)
  static const int grid_cap_env = getenv("NVBX_INTEG_GRID") ? atoi(getenv("NVBX_INTEG_GRID")) : 0;    // (env: tools/integ_grid_sweep.sh, tools/fused_grid_sweep.sh)
  const int grid_cap = grid_cap_env > 0 ? grid_cap_env : (NB > 1 ? 512 : 1024);
  // ... or fewer when the view is smaller: sized from the view count of the last launch the GPU has finished (pinned host memory, not
  // waited for) + 25 % + 64; a hint only -- the kernel grid-strides over whatever the count turns out to be
  // (no launch finished yet -- a new or just cleared map: the full grid; sized for 64 blocks, the first scans of a LiDAR map, enqueued faster
  //  than the first finishes, took 1.7 ms each for their 112 k blocks: the whole of round 4's first "exploring" LiDAR figure)
  // (the margin over the last count, NVBX_GRID_MARGIN="percent,blocks": A/B only -- tools/env_ab.sh; 25 % + 64 is what a view that grows while exploring needs)
  static const int mg_pct = getenv("NVBX_GRID_MARGIN") ? atoi(getenv("NVBX_GRID_MARGIN")) : 25;
  static const int mg_abs = (getenv("NVBX_GRID_MARGIN") && strchr(getenv("NVBX_GRID_MARGIN"), ',')) ? atoi(strchr(getenv("NVBX_GRID_MARGIN"), ',') + 1) : 64;
  const int64_t n_hint = std::max<int64_t>(0, __atomic_load_n(&m->h_mirror[2], __ATOMIC_RELAXED));
  const int64_t n_want = n_hint + n_hint * mg_pct / 100 + mg_abs;
  const int64_t want = n_hint == 0 ? (int64_t)grid_cap : ((n_want + 7) / 8) * 8;
  st.grid = (int)std::max<int64_t>(8, std::min<int64_t>(std::min<int64_t>(m->capacity, grid_cap), want));
  st.spec_lanes = (int32_t)std::min<int64_t>(64, (n_want + st.grid - 1) / st.grid);
  return NVBX_OK;
}
// step 3 (fused form): the riders of the TSDF-update launch
// [distance transform the held-back updateEsdf has just armed][TSDF update of this frame][colour integration of the held-back frame]
template <int NB>
static void depth_step_fused_riders(nvbx_mapper* m, DepthSteps<NB>& st) {
  st.n_edt = 0; st.ea_edt = m->edt_args;
  static const int edt_riders = getenv("NVBX_EDT_RIDERS") ? atoi(getenv("NVBX_EDT_RIDERS")) : 256;      // (A/B; a multiple of 8)
  if (m->edt_pending) { st.n_edt = edt_riders; m->edt_pending = false; }
  st.cand = st.tr.cand;
  st.cand_idx = st.tr.cand_cnt_idx;
  const int64_t c_hint = std::max<int64_t>(0, __atomic_load_n(&m->h_mirror[3], __ATOMIC_RELAXED));         // candidates of the last colour frame the GPU has finished
  // (no colour frame: update + distance transform only; no colour launch finished yet -- a new or just cleared map: as many as the TSDF part)
  static const int color_cap = getenv("NVBX_COLOR_GRID") ? atoi(getenv("NVBX_COLOR_GRID")) : 1024;      // (A/B: workgroups of the colour part, tools/fused_grid_sweep.sh)
  static const int mg_pct = getenv("NVBX_GRID_MARGIN") ? atoi(getenv("NVBX_GRID_MARGIN")) : 25;       // (depth_step_between)
  static const int mg_abs = (getenv("NVBX_GRID_MARGIN") && strchr(getenv("NVBX_GRID_MARGIN"), ',')) ? atoi(strchr(getenv("NVBX_GRID_MARGIN"), ',') + 1) : 64;
  st.cgrid = !st.has_color ? 0 : (int)std::max<int64_t>(8, std::min<int64_t>(std::min<int64_t>(m->capacity, color_cap), c_hint == 0 ? (int64_t)st.grid : ((c_hint + c_hint * mg_pct / 100 + mg_abs + 7) / 8) * 8));
  // a held-back union step of the multi-GPU exchange (nvbx_mark_esdf_dirty_gathered_deferred) rides here in eight workgroups: the peers'
  // blocks become ESDF-dirty for the NEXT marking pass (its own marking launch, or a ride in the colour launch, would be a third launch;
  // beside this frame's view marking it would meet blocks that launch is just allocating -- DESIGN.md 6.1)
  st.imp = ImportArgs{};
  if (m->import_pending) {
    st.imp.g = m->import_ptr; st.imp.world = m->import_world; st.imp.self_rank = m->import_rank; st.imp.max_count = m->import_max; st.imp.n_wg = 8;
    m->import_pending = false;
  }
}
// step 4: behind the TSDF-update launch
template <typename Sensor>
static int depth_step_after(nvbx_mapper* m, int n_frames) {
  NVBX_HIP(hipGetLastError());
  m->last_view_frame = m->frame_id;
  if (!Sensor::kLongRays) { m->last_camera_view_frame = m->frame_id; m->last_camera_view_mask = 1u << (n_frames - 1); }   // (a batch: the LAST camera's view, as separate calls would leave it)
  m->last_view_batch = n_frames;
  if (m->p.projective_layer_type == 2 && m->update_freespace()) return NVBX_E_DEVICE;     // TSDF with freespace (dynamic mapping)
  return m->mark_main();
}

template <typename Img, typename Sensor, int NB>
static int integrate_depth_impl(nvbx_mapper* m, FrameSet<Img, NB> fs, const Sensor& sensor) {
  // (frames of a held-back colour image this call carries out: let go of on every way out, behind the launches that read them)
  struct ReleaseFrames { nvbx_mapper* m; ~ReleaseFrames() { m->release_consumed_frames(); } } release_frames{m};
  DepthSteps<NB> st;
  { const int rc = depth_step_before_mark_view<Img, Sensor, NB>(m, fs, st); if (rc) return rc; }
  bool grid_view = false;
  { const int rc = launch_view_grid(m, fs, sensor, st.tiles, st.tr.fence_report, &grid_view); if (rc) return rc; }
  if (!grid_view)
  NVBX_LAUNCH_SMEM(m, (k_mark_view<Img, Sensor, NB>), dim3(st.tiles + st.edt_wg + st.tr.n_wg + st.tr.n_scan_wg + st.tr.n_mark_wg), dim3(Sensor::kThreads), mark_view_smem<Sensor>(st.edt_wg > 0), m->d, fs, sensor, (int4*)m->view_list, (int32_t)m->capacity,
              (int32_t)(m->premark_consumed ? 1 : 0), (int32_t)st.edt_wg, st.ea, st.tr);
  { const int rc = depth_step_between<Sensor, NB>(m, st); if (rc) return rc; }
  const int grid = st.grid; const int32_t spec_lanes = st.spec_lanes; const bool plain = st.plain;
  uint8_t* view_class = nullptr; int32_t* dense_list = nullptr;
  { const int rc = launch_lidar_sparse(m, fs, sensor, plain, &view_class, &dense_list); if (rc) return rc; }
  if (Sensor::kLongRays && m->p.projective_layer_type != 1) m->lidar_integrated = true;      // (blocks may be F_BAND_STALE from here on)
  if (st.fused) {
    if constexpr (Sensor::kRiders) {
      depth_step_fused_riders<NB>(m, st);
      const dim3 g((unsigned)(st.n_edt + grid + st.cgrid + st.imp.n_wg));
#define NVBX_FUSED_LAUNCH(PIX, PLAIN, FC) NVBX_LAUNCH(m, (k_integrate_tsdf_color<Img, PIX, NB, PLAIN>), g, dim3(512), m->d, fs, sensor, (const int4*)m->view_list, (int32_t)m->capacity, \
        m->mesh_list_live(), m->view_export, (int32_t)m->view_export_cap, spec_lanes, (int32_t)grid, FC, (const float*)m->synth, st.f_srows, st.f_scols, st.cand, st.cand_idx, st.n_edt, st.ea_edt, st.imp)
      if (NB > 1 || st.f_kind == 0) {
        if (plain) NVBX_FUSED_LAUNCH(PixRgb8, true, st.fsc); else NVBX_FUSED_LAUNCH(PixRgb8, false, st.fsc);
      } else if constexpr (NB == 1) {
        FrameSetC<PixBgra8, 1> fc; memcpy(&fc, &st.fsc, sizeof(fc));       // (one layout, color.hip static_assert)
        if (plain) NVBX_FUSED_LAUNCH(PixBgra8, true, fc); else NVBX_FUSED_LAUNCH(PixBgra8, false, fc);
      }
#undef NVBX_FUSED_LAUNCH
    }
  } else
  if (plain) NVBX_LAUNCH(m, (k_integrate_tsdf<Img, Sensor, NB, true>), dim3(grid), dim3(512), m->d, fs, sensor, (const int4*)m->view_list, (int32_t)m->capacity,
                         m->mesh_list_live(), m->view_export, (int32_t)m->view_export_cap, spec_lanes, (const uint8_t*)view_class, (const int32_t*)dense_list);
  else NVBX_LAUNCH(m, (k_integrate_tsdf<Img, Sensor, NB, false>), dim3(grid), dim3(512), m->d, fs, sensor, (const int4*)m->view_list, (int32_t)m->capacity,
                   m->mesh_list_live(), m->view_export, (int32_t)m->view_export_cap, spec_lanes, (const uint8_t*)view_class, (const int32_t*)dense_list);
  return depth_step_after<Sensor>(m, fs.n);
}

// Two mappers, one depth frame each (same image size, same stream), in TWO launches instead of four: defined as equal to integrate_depth_impl(ma) followed by
// integrate_depth_impl(mb) -- the maps share nothing, so running each launch's two halves side by side changes no result (tests/test_gpu_round6.py).  Both
// frames are camera frames that have passed integrate_cameras' own preparation (pair_prepare below).
template <typename Img>
static int integrate_depth_pair_impl(nvbx_mapper* ma, FrameSet<Img, 1> fa, nvbx_mapper* mb, FrameSet<Img, 1> fb) {
  struct ReleaseFrames { nvbx_mapper* m; ~ReleaseFrames() { m->release_consumed_frames(); } } release_a{ma}, release_b{mb};
  DepthSteps<1> sa, sb;
  { const int rc = depth_step_before_mark_view<Img, CameraSensor, 1>(ma, fa, sa); if (rc) return rc; }
  { const int rc = depth_step_before_mark_view<Img, CameraSensor, 1>(mb, fb, sb); if (rc) { ma->pipelined_order = false; return rc; } }
  // (a distance transform armed in classic order would ride with the LDS of an EdtShared: launched on its own first -- rare, a mapper that has just left the classic order)
  if (sa.edt_wg) { ma->edt_pending = true; sa.edt_wg = 0; if (ma->flush_edt()) return NVBX_E_DEVICE; }
  if (sb.edt_wg) { mb->edt_pending = true; sb.edt_wg = 0; if (mb->flush_edt()) return NVBX_E_DEVICE; }
  // (tiles-first layout inside each mapper's numbering, whether or not it has riders: the pair kernel deals the tiles of both out first)
  sa.tr.n_tile_wg = sa.tiles; sb.tr.n_tile_wg = sb.tiles;
  MarkViewArgs<Img> A{ma->d, fa, (int4*)ma->view_list, (int32_t)ma->capacity, (int32_t)(ma->premark_consumed ? 1 : 0), 0, sa.ea, sa.tr, sa.tiles, sa.tiles + sa.tr.n_wg + sa.tr.n_scan_wg + sa.tr.n_mark_wg};
  MarkViewArgs<Img> B{mb->d, fb, (int4*)mb->view_list, (int32_t)mb->capacity, (int32_t)(mb->premark_consumed ? 1 : 0), 0, sb.ea, sb.tr, sb.tiles, sb.tiles + sb.tr.n_wg + sb.tr.n_scan_wg + sb.tr.n_mark_wg};
  mb->enqueue_seq++;
  NVBX_LAUNCH_SMEM(ma, (k_mark_view_pair<Img>), dim3((unsigned)(A.n_wg + B.n_wg)), dim3(CameraSensor::kThreads), mark_view_smem<CameraSensor>(false), A, B);
  { const int rc = depth_step_between<CameraSensor, 1>(ma, sa); if (rc) { mb->pipelined_order = false; return rc; } }
  { const int rc = depth_step_between<CameraSensor, 1>(mb, sb); if (rc) return rc; }
  // the TSDF-update launch in its fused form for both (a mapper with nothing held back: no riders -- the same worker as k_integrate_tsdf)
  if (sa.fused) depth_step_fused_riders<1>(ma, sa);      // (else: zero riders, DepthSteps' defaults)
  if (sb.fused) depth_step_fused_riders<1>(mb, sb);
  // (the second mapper of a pair is the foreground mapper: a few blocks.  Its distance transform gets 64 workers instead of 256 -- they grid-stride, and the
  //  launch is residency-bound: every idle 8-wavefront workgroup holds a slot for ~1.5 us)
  static const int pair_b_edt = getenv("NVBX_PAIR_B_EDT_RIDERS") ? atoi(getenv("NVBX_PAIR_B_EDT_RIDERS")) : 64;
  if (sb.n_edt > pair_b_edt && pair_b_edt >= 8) sb.n_edt = pair_b_edt & ~7;
  const int kind = sa.has_color ? sa.f_kind : (sb.has_color ? sb.f_kind : 0);
  auto fused_args = [&](nvbx_mapper* m, const FrameSet<Img, 1>& f, const DepthSteps<1>& st, auto* out) {
    using FA = std::remove_pointer_t<decltype(out)>;
    FA x{}; x.m = m->d; x.fs = f; x.view_list = (const int4*)m->view_list; x.list_cap = (int32_t)m->capacity; x.mesh_list = m->mesh_list_live(); x.view_export = m->view_export;
    x.view_export_cap = (int32_t)m->view_export_cap; x.spec_lanes = st.spec_lanes; x.n_tsdf_wg = (int32_t)st.grid; memcpy(&x.fsc, &st.fsc, sizeof(x.fsc));
    x.synth = (const float*)m->synth; x.srows = st.f_srows; x.scols = st.f_scols; x.cand = st.cand; x.cand_cnt_idx = st.cand_idx; x.n_edt_wg = st.n_edt; x.ea = st.ea_edt; x.imp = st.imp;
    x.n_wg = st.n_edt + st.grid + st.cgrid + st.imp.n_wg;
    *out = x;
  };
  mb->enqueue_seq++;
  if (kind == 0) {
    FusedArgs<Img, PixRgb8> FA_, FB_; fused_args(ma, fa, sa, &FA_); fused_args(mb, fb, sb, &FB_);
    NVBX_LAUNCH(ma, (k_integrate_tsdf_color_pair<Img, PixRgb8>), dim3((unsigned)(FA_.n_wg + FB_.n_wg)), dim3(512), FA_, FB_);
  } else {
    FusedArgs<Img, PixBgra8> FA_, FB_; fused_args(ma, fa, sa, &FA_); fused_args(mb, fb, sb, &FB_);
    NVBX_LAUNCH(ma, (k_integrate_tsdf_color_pair<Img, PixBgra8>), dim3((unsigned)(FA_.n_wg + FB_.n_wg)), dim3(512), FA_, FB_);
  }
  { const int rc = depth_step_after<CameraSensor>(ma, 1); if (rc) return rc; }
  return depth_step_after<CameraSensor>(mb, 1);
}
// the 24-bit view frame id of Entry::stamp: before it would wrap, every stamp is reset (once per 16.7 M depth frames)
__global__ void k_reset_stamps(DMap m) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= m.mask; i += gridDim.x * blockDim.x) m.table[i].stamp = STAMP_NEVER;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m.capacity; i += gridDim.x * blockDim.x) m.slot_cam[i] = STAMP_NEVER;
  if (blockIdx.x == 0 && threadIdx.x < 4) m.counters[C_VIEW_COUNT + threadIdx.x] = 0;
}
static int next_frame_id(nvbx_mapper* m) {
  if (m->frame_id >= STAMP_FRAME_MAX) {
    if (m->join_side()) return NVBX_E_DEVICE;
    NVBX_LAUNCH(m, k_reset_stamps, dim3(1024), dim3(256), m->d);
    NVBX_HIP(hipGetLastError());
    m->frame_id = 0; m->last_view_frame = 0; m->last_camera_view_frame = 0;
  }
  m->frame_id++;
  return NVBX_OK;
}

// [U] DepthPreprocessor (do_depth_preprocessing, depth_preprocessing_num_dilations): invalid-depth regions grow by n pixels.
// One thread per pixel; the (2n+1)^2 window is read through L1/L2 (the image is 1.2 MB).  Output in metres (f32).
template <typename Img>
__global__ void k_dilate_invalid(Img in, int32_t rows, int32_t cols, int32_t n, float* out) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    bool bad = false;
    for (int dr = -n; dr <= n && !bad; dr++) {
      const int rr = r + dr; if (rr < 0 || rr >= rows) continue;
      for (int dc = -n; dc <= n; dc++) {
        const int cc = c + dc; if (cc < 0 || cc >= cols) continue;
        if (!(in(pix(rr, cc, cols)) > 0.0f)) { bad = true; break; }
      }
    }
    out[i] = bad ? 0.0f : in(i);
  }
}

// what a camera integrateDepth does before it enqueues anything of its own frame: argument check, the held-back calls it cannot carry, pool growth
template <int NB>
static int cameras_prepare(nvbx_mapper* m, int32_t n, const float* T_L_C /* n x 16 */) {
  for (int c = 0; c < n; c++)
    if (!nvbx_pose_in_range(T_L_C + 16 * c, m->p.voxel_size * 8.0f, m->p.max_integration_distance_m + 2.0f * m->p.truncation_distance_vox * m->p.voxel_size)) {
      set_error("integrate depth: T_L_C is not finite or lies outside the addressable block range (+-2^20 blocks)"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  const bool dilate_first = m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0;
  // held-back integrateColor / updateEsdf that this call cannot carry out in pipelined order are replayed NOW, with the whole held-back
  // state in view (the replayed calls launch / re-arm the held-back EDT themselves) -- before the EDT is hidden from join_side below
  const bool carry = !dilate_first && (m->color_pending.on ? ((NB == 1) == (m->color_pending.n == 1)) : m->esdf_only_carry());      // this call can carry the held-back calls out in pipelined order
  if (!carry && m->replay_deferred()) return NVBX_E_DEVICE;
  { const bool pend = m->edt_pending, ipend = m->import_pending; m->edt_pending = false; m->import_pending = false;
    // (join_side would launch a held-back EDT / union step; the EDT rides in k_mark_view instead, the union step stays held back
    //  for the next integrateColor -- it belongs to the NEXT ESDF update and touches nothing this launch reads)
    // A held-back colour frame (+ ESDF update) stays held back too when this call can carry it out in pipelined order (a single frame,
    // no dilation launch in front); otherwise join_side replays it now.
    const bool keep = carry;
    const nvbx_mapper::ColorPending cp = m->color_pending; const bool up = m->esdf_update_pending;
    if (keep) { m->color_pending.on = false; m->esdf_update_pending = false; }
    const int rc = m->join_side(); m->edt_pending = pend; m->import_pending = ipend;
    if (keep) { m->color_pending = cp; m->esdf_update_pending = up; }
    if (rc) return NVBX_E_DEVICE; }
  { const int rc = m->maybe_grow(); if (rc) return rc; }          // (before anything of this frame is enqueued)
  return NVBX_OK;
}
// n camera frames (n = 1: MultiMapper::integrateDepth; n > 1: nvbx_integrate_depth_batch) of one image size -> one launch set
template <typename Img, int NB>
static int integrate_cameras(nvbx_mapper* m, int32_t n, const Img* imgs, int32_t rows, int32_t cols, const float* T_L_C /* n x 16 */, const nvbx_camera* cameras) {
  { const int rc = cameras_prepare<NB>(m, n, T_L_C); if (rc) return rc; }
  const bool dilate = m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0;
  if (dilate && m->flush_edt()) return NVBX_E_DEVICE;   // first launch is the dilation
  { const int rc = next_frame_id(m); if (rc) return rc; }
  FrameSet<Img, NB> fs{}; fs.n = n;
  for (int c = 0; c < n; c++) { fs.f[c] = m->make_frame(T_L_C + 16 * c, cameras + c, rows, cols, m->p.raycast_subsampling_factor); fs.img[c] = imgs[c]; }
  if (dilate) {             // (single frames only: nvbx_integrate_depth_batch falls back to separate calls)
    const int64_t npx = (int64_t)rows * cols;
    if (npx > m->depth_pre_cap) {
      NVBX_HIP(hipStreamSynchronize(m->stream));
      if (m->depth_pre) NVBX_HIP(hipFree(m->depth_pre));
      m->depth_pre = nullptr; m->depth_pre_cap = 0;
      NVBX_HIP(hipMalloc(&m->depth_pre, (size_t)npx * 4));
      m->depth_pre_cap = npx;
    }
    NVBX_LAUNCH(m, (k_dilate_invalid<Img>), dim3((unsigned)std::min<int64_t>((npx + 255) / 256, 4096)), dim3(256), imgs[0], rows, cols,
                m->p.depth_preprocessing_num_dilations, m->depth_pre);
    FrameSet<DepthF32, 1> fd{}; fd.n = 1; fd.f[0] = fs.f[0]; fd.img[0] = DepthF32{m->depth_pre};
    return integrate_depth_impl<DepthF32, CameraSensor, 1>(m, fd, CameraSensor{});
  }
  return integrate_depth_impl<Img, CameraSensor, NB>(m, fs, CameraSensor{});
}

extern "C" int nvbx_integrate_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                    const nvbx_camera* camera) {
  if (!m || !depth_dev || !T_L_C || !camera || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_depth: invalid argument (image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_depth: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  const DepthF32 img{depth_dev};
  return integrate_cameras<DepthF32, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
extern "C" int nvbx_integrate_depth_u16mm(nvbx_mapper* m, const uint16_t* depth_mm_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                          const nvbx_camera* camera) {
  if (!m || !depth_mm_dev || !T_L_C || !camera || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_depth_u16mm: invalid argument (image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_depth_u16mm: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  const DepthU16mm img{depth_mm_dev};
  return integrate_cameras<DepthU16mm, 1>(m, 1, &img, rows, cols, T_L_C, camera);
}
// Up to NVBX_MAX_BATCH camera frames (same image size) in ONE launch set: see include/nvblox_hip.h
extern "C" int nvbx_integrate_depth_batch(nvbx_mapper* m, int32_t n, const float* const* depth_dev, int32_t rows, int32_t cols, const float* T_L_C,
                                          const nvbx_camera* cameras) {
  if (!m || n < 1 || n > MAX_BATCH || !depth_dev || !T_L_C || !cameras || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_depth_batch: invalid argument (1 <= n <= 8, image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  for (int c = 0; c < n; c++)
    if (!depth_dev[c] || !nvbx_camera_matches(cameras + c, rows, cols)) { set_error("nvbx_integrate_depth_batch: every camera's width/height must equal the images' cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  // what a batch cannot express falls back to the separate calls it is defined by: per-frame freespace time stamps, depth dilation
  if (n == 1 || m->p.projective_layer_type == 2 || (m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0)) {
    for (int c = 0; c < n; c++) { const int rc = nvbx_integrate_depth(m, depth_dev[c], rows, cols, T_L_C + 16 * c, cameras + c); if (rc) return rc; }
    return NVBX_OK;
  }
  DepthF32 imgs[MAX_BATCH];
  for (int c = 0; c < n; c++) imgs[c] = DepthF32{depth_dev[c]};
  return integrate_cameras<DepthF32, MAX_BATCH>(m, n, imgs, rows, cols, T_L_C, cameras);
}

// One depth frame each for TWO mappers on one stream -- MultiMapper::integrateDepth of the dynamic and the human mapping types: the background mapper takes the
// unmasked part of the depth image, the foreground (occupancy) mapper the masked part, nvblox_node.cpp:1057-1062 -- in two launches instead of four.
// See include/nvblox_hip.h; whatever the pair cannot express falls back to the two calls it is defined by.
static bool pair_can_fuse(const nvbx_mapper* m) {
  static const int fuse_on = getenv("NVBX_FUSE_COLC") ? atoi(getenv("NVBX_FUSE_COLC")) : 1;
  if (m->use_side || m->capacity > (1ll << 24)) return false;
  if (m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0) return false;          // (a dilation launch in front)
  if (m->color_pending.on && (m->color_pending.n != 1 || !fuse_on || !fused_colour_applies(m))) return false;      // (its colour frame would be carried in three launches)
  return true;
}
extern "C" int nvbx_integrate_depth_pair(nvbx_mapper* ma, const float* depth_a_dev, nvbx_mapper* mb, const float* depth_b_dev, int32_t rows, int32_t cols,
                                         const float T_L_C[16], const nvbx_camera* camera) {
  if (!ma || !mb || ma == mb || !depth_a_dev || !depth_b_dev || !T_L_C || !camera || !image_dims_ok(rows, cols)) { set_error("nvbx_integrate_depth_pair: invalid argument (two different mappers, image sides 1 .. 32768)"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_integrate_depth_pair: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  static const int pair_on = getenv("NVBX_DEPTH_PAIR") ? atoi(getenv("NVBX_DEPTH_PAIR")) : 1;       // (A/B: 0 = always the two separate calls)
  if (!pair_on || ma->device != mb->device || ma->stream != mb->stream || !pair_can_fuse(ma) || !pair_can_fuse(mb)) {
    const int rc = nvbx_integrate_depth(ma, depth_a_dev, rows, cols, T_L_C, camera); if (rc) return rc;
    return nvbx_integrate_depth(mb, depth_b_dev, rows, cols, T_L_C, camera);
  }
  // each mapper's own preparation, in call order (held-back calls it cannot carry are replayed, pools grow), then the frame ids
  { const int rc = cameras_prepare<1>(ma, 1, T_L_C); if (rc) return rc; }
  { const int rc = cameras_prepare<1>(mb, 1, T_L_C); if (rc) return rc; }
  { const int rc = next_frame_id(ma); if (rc) return rc; }
  { const int rc = next_frame_id(mb); if (rc) return rc; }
  FrameSet<DepthF32, 1> fa{}, fb{}; fa.n = 1; fb.n = 1;
  fa.f[0] = ma->make_frame(T_L_C, camera, rows, cols, ma->p.raycast_subsampling_factor); fa.img[0] = DepthF32{depth_a_dev};
  fb.f[0] = mb->make_frame(T_L_C, camera, rows, cols, mb->p.raycast_subsampling_factor); fb.img[0] = DepthF32{depth_b_dev};
  return integrate_depth_pair_impl<DepthF32>(ma, fa, mb, fb);
}

// ------------------------------------------------------------------------------------------------ multi-GPU: measurement exchange
// SURVEY.md 8e option (B), made exact.  One camera per GPU; what overlapping cameras must agree on is the TSDF.  Exchanging fused
// {distance, weight} blocks and re-fusing them on an owner is only approximately the sequential result (the weight clamp and the
// distance clamp do not commute with a weighted mean).  Exchanging MEASUREMENTS is exact: rank r runs the view calculation and the
// projection / depth sampling of ITS camera -- the expensive, sharded part -- and emits, per block in view, the 512 per-voxel pairs
// {measured depth ds, voxel depth vd} (a 4 KiB payload, the size of a TSDF block); the buffers are all-gathered (RCCL over xGMI); every
// rank then applies every camera's measurements to its map in RANK ORDER with the same per-voxel update the integrator uses.  Result:
// every rank holds the SAME map, bit-identical to one mapper integrating the cameras in rank order (nvbx_integrate_depth_batch) --
// or, with an owner filter (owner = Index3DHash(block) mod owner_mod), its shard of that map.  Two launches per apply for any G.
struct MeasRec { int32_t x, y, z, rank; float2 v[512]; };       // == nvbx_measurement_block (4112 B)
static_assert(sizeof(MeasRec) == sizeof(nvbx_measurement_block), "measurement record layout");

template <typename Img>
__global__ __launch_bounds__(512) void k_measure_tsdf(DMap m, Frame f, Img depth, CameraSensor sensor, const int4* view_list, int32_t list_cap,
                                                      MeasRec* out, int32_t* out_count, int32_t out_cap) {
  int32_t n = m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  if (n > list_cap) n = list_cap;
  if (n > out_cap) n = out_cap;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) *out_count = n;
  if (blockIdx.x == 0 && tid == 64) __hip_atomic_store(&m.host_mirror[0], m.counters[C_FREE_TOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int vx = tid >> 6, vy = (tid >> 3) & 7, vz = tid & 7;
  for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int4 rec = view_list[i];
    float pc[3], org[3], off[3];             // (the voxel centre exactly as k_integrate_tsdf evaluates it: the fused map is bit-identical)
    sensor_block_origin(f, rec.y, rec.z, rec.w, org); sensor_voxel_offset(f, vx, vy, vz, off);
    pc[0] = org[0] + off[0]; pc[1] = org[1] + off[1]; pc[2] = org[2] + off[2];
    float ds = 0.0f, vd = 0.0f;
    const int got = sensor.sample(f, depth, pc, &ds, &vd);
    // {ds, vd}: vd < 0 = the voxel is not touched; ds < 0 = it projects onto invalid depth (invalid_depth_decay); else a measurement
    float2 o = make_float2(0.0f, -1.0f);
    if (got > 0) o = make_float2(ds, vd); else if (got < 0) o = make_float2(-1.0f, vd);
    MeasRec* r = out + i;
    if (tid == 0) { r->x = rec.y; r->y = rec.z; r->z = rec.w; r->rank = 0; }
    if (tid == 64 && slot_ok((uint32_t)rec.x)) m.slot_cam[rec.x] = (f.frame_id << 8) | 1u;
    r->v[tid] = o;
  }
}
// pass 1 of an apply: one thread per record of every rank -- block lookup / allocation, position table, union list (= the view list)
__global__ void k_apply_index(DMap m, const MeasRec* all, const int32_t* counts, int32_t world, int64_t stride, int32_t owner_mod, int32_t owner_rank,
                              uint32_t frame_id, int32_t* postab, int4* view_list, int32_t list_cap) {
  int32_t* cnt = &m.counters[C_VIEW_COUNT + (frame_id & 3)];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) m.counters[C_VIEW_COUNT + ((frame_id + 1) & 3)] = 0;
  const int32_t r = (int32_t)blockIdx.y;
  int64_t n = counts[r]; if (n > stride) n = stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const MeasRec* rec = all + (size_t)r * stride + i;
    const int32_t x = rec->x, y = rec->y, z = rec->z;
    if (owner_mod > 1 && (int32_t)(index_hash(x, y, z) % (uint32_t)owner_mod) != owner_rank) continue;
    int4 out;
    int32_t h = -1;
    const bool first = mark_block(m, pack_key(x, y, z), frame_id, 1u << r, &out, &h);
    uint32_t slot;
    if (first) slot = (uint32_t)out.x;
    else { slot = SLOT_INVALID; if (h >= 0) { do { slot = ld_slot_acquire(&m.table[h]); } while (slot == SLOT_INVALID); } }   // (the entry mark_block itself reached: no second lookup)
    if (!slot_ok(slot)) continue;                      // pool exhausted
    postab[(size_t)slot * MAX_BATCH + r] = (int32_t)i + 1;
    if (first) { const int32_t p = atomicAdd(cnt, 1); if (p < list_cap) view_list[p] = out; }
  }
}
// pass 2: one workgroup per block of the union; the ranks' measurements are applied to each voxel in rank order, in registers
__global__ __launch_bounds__(512) void k_apply_fuse(DMap m, Frame f, const MeasRec* all, int64_t stride, int32_t world, int32_t* postab,
                                                    const int4* view_list, int32_t list_cap, int32_t mesh_list) {
  int32_t n = m.counters[C_VIEW_COUNT + (f.frame_id & 3)];
  if (n > list_cap) n = list_cap;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 64) __hip_atomic_store(&m.host_mirror[0], m.counters[C_FREE_TOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t slot = (uint32_t)view_list[i].x;
    if (!slot_ok(slot)) continue;
    float2* vp = &m.tsdf[(size_t)slot * 512 + tid];
    float2 fin = *vp;
    uint32_t old = 0;
    if (tid == 0) old = atomicOr(&m.slot_flags[slot], F_TSDF | F_DIRTY_ESDF | F_DIRTY_MESH);
    bool touched = false;
    uint32_t cams = 0u;
    for (int r = 0; r < world; r++) {
      const int32_t p = postab[(size_t)slot * MAX_BATCH + r];      // uniform
      if (!p) continue;
      cams |= 1u << r;
      const float2 mv = all[(size_t)r * stride + (p - 1)].v[tid];
      if (mv.y < 0.0f) continue;
      if (f.occupancy) { if (!(mv.x < 0.0f)) { fin = make_float2(occupancy_update(f, fin.x, mv.x, mv.y), 0.0f); touched = true; } }
      else if (mv.x < 0.0f) { if (f.invalid_decay >= 0.0f) { fin = make_float2(fin.x, fin.y * f.invalid_decay); touched = true; } }
      else if (tsdf_fuse(f, &fin, mv.x, mv.y)) touched = true;
    }
    if (touched) *vp = fin;
    if (tid == 64) m.slot_cam[slot] = (f.frame_id << 8) | cams;
    __syncthreads();                                                  // every lane has read the table before it is cleared
    if (tid < MAX_BATCH) postab[(size_t)slot * MAX_BATCH + tid] = 0;
    if (!f.occupancy) {
      const int any_band = __syncthreads_or(in_band(fin.x, fin.y, f.trunc) ? 1 : 0);
      if (tid == 0) { if (any_band) atomicOr(&m.slot_flags[slot], F_BAND); else atomicAnd(&m.slot_flags[slot], ~F_BAND); if (old & F_BAND_STALE) atomicAnd(&m.slot_flags[slot], ~F_BAND_STALE); }
    }
    if (tid == 0) {
      if (!(old & F_DIRTY_ESDF)) list_append(m, S_LIST_ESDF_DIRTY, (int32_t)slot);
      if (!(old & F_DIRTY_MESH)) list_append(m, mesh_list, (int32_t)slot);
    }
  }
}

extern "C" int nvbx_measure_depth(nvbx_mapper* m, const float* depth_dev, int32_t rows, int32_t cols, const float T_L_C[16], const nvbx_camera* camera,
                                  nvbx_measurement_block* out_dev, int32_t* count_dev, int64_t capacity_blocks) {
  if (!m || !depth_dev || !T_L_C || !camera || !out_dev || !count_dev || capacity_blocks <= 0 || !image_dims_ok(rows, cols)) { set_error("nvbx_measure_depth: invalid argument"); return NVBX_E_INVALID; }
  if (!nvbx_camera_matches(camera, rows, cols)) { set_error("nvbx_measure_depth: camera width/height must equal the image's cols/rows, focal lengths > 0"); return NVBX_E_INVALID; }
  if (!nvbx_pose_in_range(T_L_C, m->p.voxel_size * 8.0f, m->p.max_integration_distance_m + 2.0f * m->p.truncation_distance_vox * m->p.voxel_size)) {
    set_error("nvbx_measure_depth: T_L_C is not finite or lies outside the addressable block range"); return NVBX_E_INVALID; }
  // measure + apply is DEFINED as equal to nvbx_integrate_depth_batch; what the batch cannot express either (depth dilation, per-frame
  // freespace time stamps) is refused here instead of silently measured without it
  if (m->p.projective_layer_type == 2 || (m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0)) {
    set_error("nvbx_measure_depth: mappers with depth preprocessing (dilation) or a freespace layer integrate per frame -- use nvbx_integrate_depth"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  { const int rc = m->maybe_grow(); if (rc) return rc; }
  { const int rc = next_frame_id(m); if (rc) return rc; }
  FrameSet<DepthF32, 1> fs{}; fs.n = 1; fs.img[0] = DepthF32{depth_dev};
  fs.f[0] = m->make_frame(T_L_C, camera, rows, cols, m->p.raycast_subsampling_factor);
  const int s = fs.f[0].subsample;
  fs.f[0].n_ray_rows = (rows + s - 1 + s - 1) / s; fs.f[0].n_ray_cols = (cols + s - 1 + s - 1) / s; fs.f[0].cam_bit = 1u;
  const Frame& f = fs.f[0];
  // the view calculation against the local map: blocks in view are looked up / allocated exactly as integrateDepth would (they receive
  // their values when the gathered measurements are applied)
  TraceRider no_riders{}; no_riders.fence_report = m->next_fence_report();
  NVBX_LAUNCH_SMEM(m, (k_mark_view<DepthF32, CameraSensor, 1>), dim3(mark_view_tile_wgs<CameraSensor>(f)), dim3(CameraSensor::kThreads), mark_view_smem<CameraSensor>(false), m->d, fs, CameraSensor{},
              (int4*)m->view_list, (int32_t)m->capacity, (int32_t)(m->premark_consumed ? 1 : 0), (int32_t)0, m->edt_args, no_riders);
  m->premark_consumed = false;
  NVBX_LAUNCH(m, (k_measure_tsdf<DepthF32>), dim3((unsigned)std::min<int64_t>(m->capacity, 1024)), dim3(512), m->d, f, DepthF32{depth_dev}, CameraSensor{},
              (const int4*)m->view_list, (int32_t)m->capacity, reinterpret_cast<MeasRec*>(out_dev), count_dev, (int32_t)std::min<int64_t>(capacity_blocks, INT32_MAX));
  NVBX_HIP(hipGetLastError());
  m->last_view_frame = m->frame_id; m->last_camera_view_frame = m->frame_id; m->last_camera_view_mask = 1u; m->last_view_batch = 1;
  return m->mark_main();
}

extern "C" int nvbx_apply_measurements(nvbx_mapper* m, const nvbx_measurement_block* gathered_dev, const int32_t* counts_dev, int32_t world, int64_t stride_blocks,
                                       int32_t owner_mod, int32_t owner_rank) {
  if (!m || !gathered_dev || !counts_dev || world < 1 || world > MAX_BATCH || stride_blocks <= 0 || owner_mod < 0 || (owner_mod > 1 && (owner_rank < 0 || owner_rank >= owner_mod))) {
    set_error("nvbx_apply_measurements: invalid argument (1 <= world <= 8)"); return NVBX_E_INVALID; }
  if (m->p.projective_layer_type == 2) { set_error("nvbx_apply_measurements: mappers with a freespace layer integrate per frame (time stamps)"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  { const int rc = m->maybe_grow(); if (rc) return rc; }
  if (!m->apply_postab || m->apply_postab_cap < m->capacity) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->apply_postab) NVBX_HIP(hipFree(m->apply_postab));
    m->apply_postab = nullptr; m->apply_postab_cap = 0;
    NVBX_HIP(hipMalloc(&m->apply_postab, (size_t)m->capacity * MAX_BATCH * 4));
    NVBX_HIP(hipMemsetAsync(m->apply_postab, 0, (size_t)m->capacity * MAX_BATCH * 4, m->stream));
    m->apply_postab_cap = m->capacity;
  }
  if (m->begin_dirtying()) return NVBX_E_DEVICE;
  { const int rc = next_frame_id(m); if (rc) return rc; }
  nvbx_camera none{1.f, 1.f, 0.f, 0.f, 1, 1};
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const Frame f = m->make_frame(I, &none, 1, 1, 1);          // (the integrator parameters; no camera is involved in applying measurements)
  const MeasRec* all = reinterpret_cast<const MeasRec*>(gathered_dev);
  NVBX_LAUNCH(m, k_apply_index, dim3(64, (unsigned)world), dim3(256), m->d, all, counts_dev, world, stride_blocks, owner_mod, owner_rank, m->frame_id, m->apply_postab,
              (int4*)m->view_list, (int32_t)m->capacity);
  NVBX_LAUNCH(m, k_apply_fuse, dim3((unsigned)std::min<int64_t>(m->capacity, 1024)), dim3(512), m->d, f, all, stride_blocks, world, m->apply_postab,
              (const int4*)m->view_list, (int32_t)m->capacity, m->mesh_list_live());
  NVBX_HIP(hipGetLastError());
  m->last_view_frame = m->frame_id; m->last_camera_view_frame = m->frame_id; m->last_camera_view_mask = 1u << (world - 1); m->last_view_batch = world;
  return m->mark_main();
}

// ------------------------------------------------------------------------------------------------ LiDAR
static bool same_lidar(const nvbx_lidar& a, const nvbx_lidar& b) { return memcmp(&a, &b, sizeof(a)) == 0; }

// beam direction tables: sin / cos evaluated in double on the host from the float model parameters, rounded to float
// (the oracle builds the same tables the same way, so view rays are bit-identical)
static int ensure_lidar_tables(nvbx_mapper* m, const nvbx_lidar* ld, const nvbx_lidar_model& l) {
  if (m->lidar_tab && same_lidar(m->lidar_cached, *ld)) return NVBX_OK;
  const size_t n = (size_t)l.rows + (size_t)l.cols;
  if (n > m->lidar_tab_cap) {
    NVBX_HIP(hipStreamSynchronize(m->stream));
    if (m->lidar_tab) NVBX_HIP(hipFree(m->lidar_tab));
    m->lidar_tab = nullptr; m->lidar_tab_cap = 0;
    NVBX_HIP(hipMalloc(&m->lidar_tab, n * sizeof(float2)));
    m->lidar_tab_cap = n;
  }
  m->lidar_host.resize(n * 2);
  for (int k = 0; k < l.rows; k++) {
    const double el = (double)l.max_el - (double)k * (double)l.rpp_el;
    m->lidar_host[2 * (size_t)k] = (float)sin(el); m->lidar_host[2 * (size_t)k + 1] = (float)cos(el);
  }
  for (int j = 0; j < l.cols; j++) {
    const double az = -(double)NVBX_PI_F + (double)j * (double)l.rpp_az;
    m->lidar_host[2 * ((size_t)l.rows + j)] = (float)sin(az); m->lidar_host[2 * ((size_t)l.rows + j) + 1] = (float)cos(az);
  }
  NVBX_HIP(hipMemcpyAsync(m->lidar_tab, m->lidar_host.data(), n * sizeof(float2), hipMemcpyHostToDevice, m->stream));
  NVBX_HIP(hipStreamSynchronize(m->stream));    // once per sensor model
  m->lidar_cached = *ld;
  return NVBX_OK;
}

static bool lidar_ok(const nvbx_lidar* ld) {
  return ld && ld->num_azimuth_divisions >= 2 && ld->num_elevation_divisions >= 2 && ld->max_elevation_rad > ld->min_elevation_rad &&
         image_dims_ok(ld->num_elevation_divisions, ld->num_azimuth_divisions);
}

extern "C" int nvbx_integrate_lidar_depth(nvbx_mapper* m, const float* range_dev, int32_t rows, int32_t cols, const float T_L_C[16],
                                          const nvbx_lidar* lidar) {
  if (!m || !range_dev || !T_L_C || !lidar_ok(lidar) || rows != lidar->num_elevation_divisions || cols != lidar->num_azimuth_divisions) {
    set_error("nvbx_integrate_lidar_depth: invalid argument (range image must be elevation x azimuth divisions)"); return NVBX_E_INVALID;
  }
  if (!nvbx_pose_in_range(T_L_C, m->p.voxel_size * 8.0f, m->p.lidar_max_integration_distance_m + 2.0f * m->p.truncation_distance_vox * m->p.voxel_size)) {
    set_error("nvbx_integrate_lidar_depth: T_L_C is not finite or lies outside the addressable block range (+-2^20 blocks)"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  const nvbx_lidar_model l = nvbx_lidar_make(cols, rows, lidar->min_valid_range_m, lidar->min_elevation_rad, lidar->max_elevation_rad);
  const int rc = ensure_lidar_tables(m, lidar, l); if (rc) return rc;
  { const int rcg = m->maybe_grow(); if (rcg) return rcg; }
  { const int rc2 = next_frame_id(m); if (rc2) return rc2; }
  nvbx_camera none{1.f, 1.f, 0.f, 0.f, cols, rows};
  FrameSet<DepthF32, 1> fs{}; fs.n = 1; fs.img[0] = DepthF32{range_dev};
  fs.f[0] = m->make_frame(T_L_C, &none, rows, cols, m->p.raycast_subsampling_factor);
  fs.f[0].max_dist = m->p.lidar_max_integration_distance_m;
  LidarSensor s{l, (const float2*)m->lidar_tab, (const float2*)m->lidar_tab + rows,
                m->p.lidar_linear_interpolation_max_allowable_difference_vox * m->p.voxel_size,
                m->p.lidar_nearest_interpolation_max_allowable_dist_to_ray_vox * m->p.voxel_size};
  return integrate_depth_impl<DepthF32, LidarSensor, 1>(m, fs, s);
}

// depthImageFromPointcloudKernel (conversions/pointcloud_conversions.cu:118-150): last writer wins
__global__ void k_depth_from_points(const float* pts, int64_t n, nvbx_lidar_model l, float* img) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (isnan(p[0]) || isnan(p[1]) || isnan(p[2])) continue;
    const float r = nvbx_lidar_range(p);
    float u, v;
    if (!nvbx_lidar_project(&l, p, r, &u, &v)) continue;
    const int c = (int)floorf(u), rr = (int)floorf(v);
    if (c < 0 || rr < 0 || c >= l.cols || rr >= l.rows) continue;
    img[(int64_t)rr * l.cols + c] = r;
  }
}
extern "C" int nvbx_depth_image_from_pointcloud(nvbx_mapper* m, const float* points_xyz_dev, int64_t n_points, const nvbx_lidar* lidar,
                                                float* range_dev) {
  if (!m || !points_xyz_dev || n_points < 0 || !lidar_ok(lidar) || !range_dev) { set_error("nvbx_depth_image_from_pointcloud: invalid argument"); return NVBX_E_INVALID; }
  NVBX_HIP(hipSetDevice(m->device));
  if (m->join_side()) return NVBX_E_DEVICE;
  const nvbx_lidar_model l = nvbx_lidar_make(lidar->num_azimuth_divisions, lidar->num_elevation_divisions, lidar->min_valid_range_m,
                                             lidar->min_elevation_rad, lidar->max_elevation_rad);
  NVBX_HIP(hipMemsetAsync(range_dev, 0, (size_t)l.rows * l.cols * sizeof(float), m->stream));
  if (n_points > 0)
    NVBX_LAUNCH(m, k_depth_from_points, dim3((unsigned)std::min<int64_t>((n_points + 255) / 256, 4096)), dim3(256), points_xyz_dev, n_points, l, range_dev);
  NVBX_HIP(hipGetLastError());
  return NVBX_OK;
}
