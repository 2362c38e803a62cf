// nvbx_internal.h -- device data layout + device helpers shared by the HIP kernels of libnvblox_hip.so.
//
// HBM layout (one per mapper == one per GPU), sized once for `capacity` blocks, never reallocated:
//   table      : open-addressing hash, 2^k >= 2*capacity entries of 16 B {key64, slot32, view_stamp32}.  The probe
//                start is the reference's Index3DHash x + 17191 y + 17191^2 z (nvblox_hash_utils.h:40-50) scattered by
//                a Fibonacci multiply (table_pos); linear probing; no tombstones (deallocation rebuilds the table on device).
//   slot_*     : per-slot metadata (layer/dirty flags, Index3D, back-pointer to the hash entry, ESDF epoch stamp)
//   tsdf/color : capacity x 512 x 8 B voxel pools, voxel order z + 8y + 64x (the reference's order, so block copies
//                are memcpy); one slot id addresses the TSDF, colour and ESDF block of the same Index3D.
//   esdf       : capacity x 512 x 8 B, voxel order x + 8y + 64z so that the 2-D slice plane of a block is one
//                contiguous 512-B line = one 8-byte access per lane of one wavefront; packed {f32 sq, u32 meta}.
//   site/obs/inside_bits : capacity x 8 B each, the slice plane's site / observed / inside masks of each ESDF block (one
//                __ballot word each), written by the marking pass and applied to the voxels by the distance transform
// All voxel types are 8 bytes -> every block of every layer is 4 KiB and a 512-thread workgroup (8 wave64) moves one
// block with one coalesced 8-B access per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nvbx_arith.h"

namespace nvbx {

typedef unsigned long long u64;

constexpr u64 KEY_EMPTY = ~0ull;
constexpr uint32_t SLOT_INVALID = 0xFFFFFFFFu;   // entry inserted, slot not published yet
constexpr uint32_t STAMP_NEVER = 0xFFFFFFFFu;
constexpr uint32_t SLOT_NONE = 0xFFFFFFFEu;      // entry exists but the pool was exhausted
__host__ __device__ inline bool slot_ok(uint32_t s) { return s < SLOT_NONE; }

// slot_flags bits
constexpr uint32_t F_TSDF = 1u, F_COLOR = 2u, F_ESDF = 4u, F_MESH = 8u;
constexpr uint32_t F_FREESPACE = 32u;      // (16 is the API id of the occupancy layer, which lives under F_TSDF)
constexpr uint32_t F_DIRTY_ESDF = 1u << 8, F_DIRTY_MESH = 1u << 9;
// the block was given an ESDF column by a marking pass but joins the ESDF layer (F_ESDF, layer AABB) only when the distance
// transform of that update runs: marking never changes anything the API can observe
constexpr uint32_t F_ESDF_PENDING = 1u << 10;
// on an ESDF slot: a TSDF block of this column's z band was deallocated (decay), so the column must be re-marked by the next
// ESDF update although no TSDF block of it may be dirty (or even exist); cleared by the distance transform of that update
constexpr uint32_t F_ESDF_REMARK = 1u << 11;
// on a TSDF slot: EXACTLY "some voxel of the block lies in the truncation band (weight > 1e-4 and |distance| < truncation
// distance)" -- the block-level vote of the colour integrator, kept up to date by every kernel that writes TSDF voxels
// (integration, decay, clearing, block upload; recomputed for all blocks if the truncation distance changes), so that
// integrateColor decides it from the flags it loads anyway instead of reading 4 KiB of TSDF per allocated block
// Eight bits, one per x-slab of the block = per wavefront of the 512-thread kernels that write TSDF voxels (thread = voxel
// z + 8y + 64x, wavefront w holds x = w): each wavefront owns its bit and sets / clears it from its own ballot, so the flag is
// exact without a workgroup barrier (a barrier per block cost the VALU-bound LiDAR integration 24 %).
constexpr uint32_t F_BAND_SHIFT = 12, F_BAND = 0xFFu << F_BAND_SHIFT;
// The LiDAR integration (VALU-bound, ~10^5 blocks per scan) does not keep the bits: it marks the block STALE with the flag atomic
// it issues anyway, and the first colour integration that meets a stale block votes from the TSDF once and repairs the bits.
constexpr uint32_t F_BAND_STALE = 1u << 20;
__host__ __device__ inline bool in_band(float d, float w, float trunc) { return w > 1e-4f && fabsf(d) < trunc; }
#ifdef __HIPCC__
// called by ALL lanes of a wavefront of a 512-thread block-per-workgroup kernel; `pred` = this lane's voxel is in the band
__device__ inline void publish_band(uint32_t* slot_flags, uint32_t slot, int tid, bool pred) {
  const unsigned long long b = __ballot(pred);
  if ((tid & 63) == 0) {
    const uint32_t bit = 1u << (F_BAND_SHIFT + (tid >> 6));
    if (b) atomicOr(&slot_flags[slot], bit); else atomicAnd(&slot_flags[slot], ~bit);
    if (tid == 0) atomicAnd(&slot_flags[slot], ~F_BAND_STALE);          // all eight bits are being written: exact again
  }
}
#endif

struct Entry { u64 key; uint32_t slot; uint32_t stamp; };
// Entry::stamp = (view frame id << 8) | camera mask: the depth frame (or batch of up to 8 camera frames integrated by ONE launch set,
// nvbx_integrate_depth_batch) that last had the block in view, and which cameras of that batch saw it.  Frame ids are 24 bits
// (1 .. STAMP_FRAME_MAX; the host resets every stamp when the counter would wrap); STAMP_NEVER = all ones never matches.
constexpr uint32_t STAMP_FRAME_MAX = 0xFFFFF0u;
constexpr int MAX_BATCH = 8;
constexpr int MAX_IMAGE_DIM = 32768;   // image sides accepted by the C-ABI (pixel indices then fit 31 bits, see pix())
inline bool image_dims_ok(int64_t rows, int64_t cols) { return rows > 0 && cols > 0 && rows <= MAX_IMAGE_DIM && cols <= MAX_IMAGE_DIM; }
__host__ __device__ inline uint32_t stamp_frame(uint32_t s) { return s >> 8; }

// counters[] indices (device int32 array, mirrored to pinned host memory on demand)
enum {
  C_FREE_TOP = 0,      // number of free slots on the stack
  C_HIGH_WATER = 1,    // 1 + highest slot ever handed out
  C_OVERFLOW = 2,      // sticky capacity-overflow flag
  C_VIEW_COUNT = 4,    // [4..7]  ring of per-frame view-list counts (frame & 3)
  C_ESDF3_WIN = 8,     // [8..14] 3-D ESDF: block AABB changed since the last update (min x,y,z, max x,y,z) + blocks marked
  C_ESDF_UPD = 16,     // [16..31] two parity-indexed records of 8 ints for ESDF update e (record e & 1): only
                       //   +6 (window voxels) lives here; the contended fields are sharded (S_ESDF_REC below)
  C_ESDF_AABB = 32,    // [32..35] AABB of all ESDF blocks: min_x, min_y, max_x, max_y
  C_MESH_OUT = 36,     // [36..43] two parity-indexed records of mesh update e: {+0 list entries}; the contended
                       //   fields are sharded (S_MESH_REC below)
  C_LIVE = 44,         // (unused since round 3: live blocks = capacity - C_FREE_TOP)
  C_TMP = 45,          // scratch counter (point cloud compaction etc.)
  C_CLEARED = 46,      // entries of the cleared-block list (nvbx_take_cleared_blocks)
  C_MARK_DONE = 47,    // workers of the running ESDF marking pass that have finished (a pass that empties the dirty list itself, EsdfArgs::self_reset)
  C_CAND_COUNT = 48,   // [48..49] parity-indexed: colour candidate records discovered for the fused colour + TSDF launch (nvbx_color_worker.h)
  // -DNVBX_CHECK_INVARIANTS variant of the library only (tools/build_variant.sh inv "-DNVBX_CHECK_INVARIANTS"; DESIGN.md 2.8's table made executable):
  C_INV_WRITERS = 50,  // workers that write TSDF voxels / band flags and are running right now
  C_INV_I1 = 51,       // violations of I1: a TSDF-reading rider of launch 1 (sphere tracing, colour candidates, ESDF marking) met a running TSDF writer
  C_INV_I4 = 52,       // violations of I4: the marking pass took a dirty-list entry whose slot carries no projective layer (freed and not re-issued: the flag test must have stopped it)
  C_INV_I3 = 53,       // violations of I3: a colour worker of launch 2 was handed a candidate record whose slot does not name that block
  C_NUM = 56
};

// ---- sharded counters.  A counter that every workgroup of a launch bumps serialises at ~12 ns per atomic in the
// memory-side atomic unit (276 blocks x 2 list appends = 6.6 us inside a 7 us kernel).  Every such counter therefore
// exists NSH times, one copy per 64-B line, and a workgroup uses copy blockIdx.x & 7 (= its XCD); readers sum / reduce
// the copies.  Layout of DMap::shc: [(id * NSH + shard) * SH_STRIDE + field].
constexpr int NSH = 8;
constexpr int SH_STRIDE = 16;
enum {
  S_LIST_ESDF_DIRTY = 0,   // work list: TSDF slots dirtied since the last ESDF update        (field 0 = entries)
  S_LIST_MESH_DIRTY = 1,   // [1..2] work list of mesh-update parity 0 / 1                    (field 0 = entries)
  S_LIST_COLOR = 3,        // slots updated by the last colour frame                          (field 0 = entries)
  S_ESDF_REC = 4,          // [4..5] parity-indexed ESDF update record: 0..3 dirty window min_x, min_y, max_x, max_y
                           //   (block coords), 4 columns re-marked, 5 ESDF blocks swept
  S_MESH_REC = 6,          // [6..7] parity-indexed mesh update record: 0 blocks meshed, 2..3 u64 arena cursor of this
                           //   shard's arena region = vertices (low 32) | triangles (high 32)
  S_LIDAR_SPARSE = 9,      // field 0: blocks of the current LiDAR scan updated by the beam-centric launch (reset by that scan's view marking)
  S_MARK_DONE = 8,         // field 0: workers of the running self-resetting ESDF marking pass that have finished, per shard (C_MARK_DONE counts the shards)
  S_NUM = 10
};
constexpr int N_LISTS = 4;

struct DMap {
  Entry* table; uint32_t mask; uint32_t shift;   // table size = mask + 1 = 2^(32 - shift)
  uint32_t capacity;
  uint32_t* free_stack;
  int32_t* counters;
  uint32_t* slot_flags;
  int32_t* slot_index;      // 3 ints per slot
  uint32_t* slot_entry;     // slot -> hash entry
  uint32_t* slot_stamp;     // ESDF slot: marking pass that last re-marked this column (de-duplication within a pass)
  uint32_t* slot_cam;       // TSDF slot: (view frame id << 8) | camera mask of the last CAMERA depth launch that had the block in view, written by
                            //   the TSDF update of that launch.  decayTsdfExcludeLastView<Camera> reads it: the shared Entry::stamp is re-claimed by
                            //   every later view calculation (a LiDAR scan in between would otherwise take the camera view's place)
  uint32_t* slot_consumed;  // TSDF slot: marking pass that last consumed its ESDF-dirty flag (STAMP_NEVER = none); lets a
                            //   deallocating operation take back marking passes that no distance transform has followed yet
  float2* tsdf;
  uint2* color;
  uint2* esdf;
  // freespace layer of a TSDF-with-freespace mapper (dynamic mapping): {i64 last_occupied_ms, i32 consecutive_ms, u32 bit0 =
  // high-confidence freespace, bit1 = initialised}; voxel order z + 8y + 64x; allocated when first needed, all-zero for free slots
  int4* freespace;
  u64* site_bits;           // per slot: site mask of the block's ESDF slice plane (bit x + 8y); 0 for non-ESDF slots
  u64* obs_bits;            // per slot: observed mask of the slice plane, as of the last marking pass
  u64* inside_bits;         // per slot: inside mask of the slice plane, as of the last marking pass
  int32_t* shc;             // sharded counters (S_* above)
  int32_t* lists;           // N_LISTS x NSH x capacity slot ids: list l, shard s starts at ((l * NSH + s) * capacity)
  int32_t* host_mirror;     // pinned host memory, device-mapped: [0] = free slots as of the last TSDF-update launch (pool growth, mapper.hip),
                            //   [1] = high-water mark, [2] = blocks in view of that launch (sizes the next TSDF-update grid)
};

// Per-call camera / pose / parameter bundle (kernel argument, lives in SGPRs).
// The part of a frame the colour path needs (sphere tracing, candidate discovery, colour integration): pose, intrinsics, image size and a few
// scalars -- 156 bytes.  Kept apart because a BATCH passes eight of these per argument block, and a launch that carries the frames of a depth
// batch AND of a colour batch (pipelined order) has to fit the 4 KiB kernel-argument limit.
struct FrameCore {
  float R_CL[9], t_CL[3];   // p_C = R_CL p_L + t_CL
  float R_LC[9], t_LC[3];   // p_L = R_LC p_C + t_LC
  float fu, fv, cu, cv; int32_t w, h;
  int32_t rows, cols;
  float voxel_size, block_size, trunc, max_dist, max_weight;
  float occlusion_thresh;   // [U] colour occlusion threshold (m)
  int32_t subsample;        // raycast / sphere-tracing subsampling
};
struct Frame : FrameCore {
  int32_t weighting_mode, interp_nearest;
  // [U] open-choice switches (include/nvblox_hip.h): weighting formula set, sdf == -trunc edge, clamp order
  int32_t weighting_variant, skip_at_neg_trunc, clamp_before_blend;
  float invalid_decay;      // invalid_depth_decay_factor (< 0 = off)
  // occupancy mappers (projective_layer_type 1): the projective pool holds {log_odds, 0}; log-odds updates of the three regions
  int32_t occupancy; float lo_free, lo_occupied, lo_unobserved, occ_half_width;
  int32_t ws_type; float ws_min[3], ws_max[3];   // workspace bounds of the view calculator (0 = unbounded)
  int32_t n_ray_rows, n_ray_cols;
  uint32_t frame_id;        // 24-bit view frame id (Entry::stamp)
  uint32_t cam_bit;         // 1 << index of this camera in its batch (1 for a single frame)
};

__host__ __device__ inline u64 pack_key(int32_t x, int32_t y, int32_t z) {
  const u64 B = 1ull << 20;
  return (((u64)(int64_t)x + B) & 0x1FFFFFull) | ((((u64)(int64_t)y + B) & 0x1FFFFFull) << 21) |
         ((((u64)(int64_t)z + B) & 0x1FFFFFull) << 42);
}
__host__ __device__ inline void unpack_key(u64 key, int32_t* x, int32_t* y, int32_t* z) {
  *x = (int32_t)(key & 0x1FFFFFull) - (1 << 20);
  *y = (int32_t)((key >> 21) & 0x1FFFFFull) - (1 << 20);
  *z = (int32_t)((key >> 42) & 0x1FFFFFull) - (1 << 20);
}
__host__ __device__ inline uint32_t index_hash(int32_t x, int32_t y, int32_t z) {
  // nvblox_hash_utils.h:43-48 -- uint32 wrap-around is identical to truncating the size_t sum
  return (uint32_t)x + (uint32_t)y * 17191u + (uint32_t)z * (17191u * 17191u);
}

// Probe start: the reference's Index3DHash scattered by a Fibonacci multiply.  The raw hash maps x-neighbours to
// adjacent table entries (and y/z strides alias: 17191*5 = 17191^2*3 - 16 mod 2^16), which turns linear probing into
// 20-40 dependent loads per lookup on a room-sized map; the multiply restores ~1 probe at our <= 0.5 load factor.
__host__ __device__ inline uint32_t table_pos(const DMap& m, int32_t x, int32_t y, int32_t z) {
  return (index_hash(x, y, z) * 2654435761u) >> m.shift;
}

#ifdef __HIPCC__
// per-workgroup instrumentation (tools/wg_timeline.py, -DNVBX_WG_TIMES; tsdf.hip defines the buffer): nothing in the product build
#if defined(NVBX_WG_TIMES) && defined(NVBX_WGT_HERE)       // (tsdf.hip only: the buffer pointer is a device global of that translation unit)
extern __device__ unsigned long long* g_wgt;
#define NVBX_TV(k, i, v) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0 && g_wgt && blockIdx.x < 8192) g_wgt[((size_t)(k) * 8192 + blockIdx.x) * 8 + (i)] = (unsigned long long)(v); } while (0)
#else
#define NVBX_TV(k, i, v) do { } while (0)
#endif
// ---- executable invariants (the -DNVBX_CHECK_INVARIANTS variant; the product build compiles every macro to nothing)
#ifdef NVBX_CHECK_INVARIANTS
// a worker that writes TSDF voxels or band flags: bracketed by these (one lane per workgroup counts)
#define NVBX_INV_WRITER_BEGIN(m) do { __syncthreads(); if (threadIdx.x == 0) atomicAdd(&(m).counters[C_INV_WRITERS], 1); } while (0)
#define NVBX_INV_WRITER_END(m) do { __syncthreads(); if (threadIdx.x == 0) atomicSub(&(m).counters[C_INV_WRITERS], 1); } while (0)
// a rider that READS the TSDF as the last update left it: no writer may be running (checked where the rider starts and where it ends)
#define NVBX_INV_TSDF_READER(m) do { if ((threadIdx.x & 63) == 0 && __hip_atomic_load(&(m).counters[C_INV_WRITERS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) atomicAdd(&(m).counters[C_INV_I1], 1); } while (0)
#define NVBX_INV_COUNT(m, idx, cond) do { if (cond) atomicAdd(&(m).counters[idx], 1); } while (0)
#else
#define NVBX_INV_WRITER_BEGIN(m) do { } while (0)
#define NVBX_INV_WRITER_END(m) do { } while (0)
#define NVBX_INV_TSDF_READER(m) do { } while (0)
#define NVBX_INV_COUNT(m, idx, cond) do { } while (0)
#endif
__device__ inline int32_t* shc_at(const DMap& m, int id, int shard, int field) { return &m.shc[(id * NSH + shard) * SH_STRIDE + field]; }
__device__ inline int my_shard() { return (int)(blockIdx.x & (NSH - 1)); }

// append to a sharded work list (any thread; the counter is the caller's XCD copy)
__device__ inline void list_append(const DMap& m, int list, int32_t v) {
  const int sh = my_shard();
  const int32_t p = atomicAdd(shc_at(m, list, sh, 0), 1);
  if (p < (int32_t)m.capacity) m.lists[((size_t)list * NSH + sh) * m.capacity + p] = v;
}
// reader side: prefix of the shard counts, then item i of the concatenation
struct ListView { int32_t pre[NSH + 1]; };
__device__ inline int32_t list_open(const DMap& m, int list, ListView* v) {
  int32_t c[NSH];
#pragma unroll
  for (int s = 0; s < NSH; s++) c[s] = *shc_at(m, list, s, 0);         // 8 independent loads
  v->pre[0] = 0;
#pragma unroll
  for (int s = 0; s < NSH; s++) v->pre[s + 1] = v->pre[s] + min(c[s], (int32_t)m.capacity);
  return v->pre[NSH];
}
__device__ inline int32_t list_at(const DMap& m, int list, const ListView& v, int32_t i) {
  int s = 0; int32_t base = 0;                 // (selects only: a dynamically indexed pre[] would live in scratch memory)
#pragma unroll
  for (int q = 1; q < NSH; q++) if (i >= v.pre[q]) { s = q; base = v.pre[q]; }
  return m.lists[((size_t)list * NSH + s) * m.capacity + (i - base)];
}
__device__ inline void list_reset(const DMap& m, int list) {            // one thread
#pragma unroll
  for (int s = 0; s < NSH; s++) *shc_at(m, list, s, 0) = 0;
}

#include "nvbx_numbering.h"      // xcd_chunked: XCD-affine numbering of a work list's records (round 6)
static_assert(NVBX_N_XCD == NSH, "one shard per XCD");

__device__ inline int32_t floor_div8(int32_t v) { return v >> 3; }
__device__ inline int32_t mod8(int32_t v) { return v & 7; }

__device__ inline uint32_t ld_slot_acquire(const Entry* e) {
  return __hip_atomic_load(&e->slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lookup of a key inserted by an EARLIER kernel (or earlier in this kernel by this thread). Returns entry or -1.
__device__ inline int32_t hash_find(const DMap& m, int32_t x, int32_t y, int32_t z) {
  const u64 key = pack_key(x, y, z);
  uint32_t h = table_pos(m, x, y, z);
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    const u64 k = m.table[h].key;
    if (k == key) return (int32_t)h;
    if (k == KEY_EMPTY) return -1;
    h = (h + 1) & m.mask;
  }
  return -1;
}
// slot of an existing block carrying `layer`, or SLOT_NONE
__device__ inline uint32_t find_slot(const DMap& m, int32_t x, int32_t y, int32_t z, uint32_t layer) {
  const int32_t h = hash_find(m, x, y, z);
  if (h < 0) return SLOT_NONE;
  const uint32_t s = m.table[h].slot;
  if (!slot_ok(s)) return SLOT_NONE;
  return (m.slot_flags[s] & layer) ? s : SLOT_NONE;
}

// Lookups WITHOUT a layer-flag check (one 16-B entry load per probe = key + slot + stamp).  Valid wherever the pool of a
// slot that lacks the layer reads as "nothing there": the TSDF and colour pools and site_bits of such slots are all-zero
// (zeroed on free, never written otherwise), and weight 0 means unobserved / uncoloured exactly like a missing block.
__device__ inline uint4 ld_entry(const DMap& m, uint32_t h) { return *reinterpret_cast<const uint4*>(&m.table[h]); }
// finish a lookup whose first probe `e` at `h` is already loaded: slot of any block with `key`, or SLOT_NONE
__device__ inline uint32_t resolve_any(const DMap& m, u64 key, uint32_t h, uint4 e) {
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    const u64 k = ((u64)e.y << 32) | (u64)e.x;
    if (k == key) return slot_ok(e.z) ? e.z : SLOT_NONE;
    if (k == KEY_EMPTY) return SLOT_NONE;
    h = (h + 1) & m.mask;
    e = ld_entry(m, h);
  }
  return SLOT_NONE;
}
__device__ inline uint32_t any_slot(const DMap& m, int32_t x, int32_t y, int32_t z) {
  const uint32_t h = table_pos(m, x, y, z);
  return resolve_any(m, pack_key(x, y, z), h, ld_entry(m, h));
}

// Insert-if-absent. Device-scope CAS on the key decides the winner; the winner pops a pool slot and publishes it with
// an agent-scope store.  `is_new` tells the caller it won.  Returns the entry index or -1 (table full).
__device__ inline int32_t hash_insert(const DMap& m, int32_t x, int32_t y, int32_t z, uint32_t layer_flags, bool* is_new) {
  const u64 key = pack_key(x, y, z);
  uint32_t h = table_pos(m, x, y, z);
  *is_new = false;
  for (uint32_t probe = 0; probe <= m.mask; ++probe) {
    u64 k = m.table[h].key;              // may be a stale EMPTY: the CAS below is the truth
    if (k == KEY_EMPTY) {
      k = atomicCAS(&m.table[h].key, KEY_EMPTY, key);
      if (k == KEY_EMPTY) {
        *is_new = true;
        const int32_t top = atomicSub(&m.counters[C_FREE_TOP], 1);
        uint32_t slot = SLOT_NONE;
        if (top > 0) {
          slot = m.free_stack[top - 1];
          m.slot_index[3 * slot] = x; m.slot_index[3 * slot + 1] = y; m.slot_index[3 * slot + 2] = z;
          m.slot_entry[slot] = h;
          atomicOr(&m.slot_flags[slot], layer_flags);
          atomicMax(&m.counters[C_HIGH_WATER], (int32_t)slot + 1);
          // (no live-block counter: every slot is either on the free stack or live, so live = capacity - C_FREE_TOP; a third same-address
          //  atomic per new block was ~12 us of serialised atomics on a frame that allocates 1000 blocks)
        } else {
          atomicAdd(&m.counters[C_FREE_TOP], 1);
          atomicExch(&m.counters[C_OVERFLOW], 1);
        }
        __hip_atomic_store(&m.table[h].slot, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int32_t)h;
      }
    }
    if (k == key) return (int32_t)h;
    h = (h + 1) & m.mask;
  }
  atomicExch(&m.counters[C_OVERFLOW], 1);
  return -1;
}

// ---------------------------------------------------------------- numerical contract: fixed evaluation order,
// no contraction (-ffp-contract=off), IEEE division and sqrt.
__device__ inline void apply_rt(const float* R, const float* t, float x, float y, float z, float* o) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = R[3 * i + 0] * x;
    s = s + R[3 * i + 1] * y;
    s = s + R[3 * i + 2] * z;
    o[i] = s + t[i];
  }
}
__device__ inline void rotate(const float* R, float x, float y, float z, float* o) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = R[3 * i + 0] * x;
    s = s + R[3 * i + 1] * y;
    s = s + R[3 * i + 2] * z;
    o[i] = s;
  }
}
// The integrators' voxel centre in the SENSOR frame = (block origin, transformed once per block) + (the voxel's offset inside the
// block, rotated once per voxel position): sensor_block_origin() + sensor_voxel_offset().  Same value as transforming the centre
// itself up to rounding, but the per-block part is wave-uniform and the per-voxel part loop-invariant, so a voxel costs three
// additions instead of a 3 x 3 transform (the oracle evaluates the same two terms: voxel_in_sensor, oracle/nvblox_oracle.c).
__device__ inline void sensor_block_origin(const Frame& f, int32_t bx, int32_t by, int32_t bz, float* o) {
  apply_rt(f.R_CL, f.t_CL, (float)bx * f.block_size, (float)by * f.block_size, (float)bz * f.block_size, o);
}
__device__ inline void sensor_voxel_offset(const Frame& f, int vx, int vy, int vz, float* o) {
  const float h = f.voxel_size * 0.5f;
  rotate(f.R_CL, (float)vx * f.voxel_size + h, (float)vy * f.voxel_size + h, (float)vz * f.voxel_size + h, o);
}
__device__ inline float voxel_center(int32_t bi, int32_t vi, float bs, float vs) {
  return ((float)bi * bs + (float)vi * vs) + vs * 0.5f;   // layer_publishing.cpp:527
}
__device__ inline bool cam_project(const FrameCore& f, const float* p, float* u, float* v) {
  if (p[2] <= 0.0f) return false;
  *u = f.fu * NVBX_DIV(p[0], p[2]) + f.cu;          // (NVBX_DIV: the IEEE quotient, nvbx_arith.h)
  *v = f.fv * NVBX_DIV(p[1], p[2]) + f.cv;
  return *u >= 0.0f && *v >= 0.0f && *u <= (float)f.w && *v <= (float)f.h;      // (written so that a NaN is outside the image)
}

// Pixel index r * cols + c.  Image sides are at most MAX_IMAGE_DIM (checked at the C-ABI: image_dims_ok), so the full-rate 24-bit
// multiply is exact and the index fits 31 bits -- the 64-bit multiply-add the compiler emits for int64 index math is quarter rate,
// and the integrators compute four of them per voxel.
__device__ inline int32_t pix(int r, int c, int cols) { return __mul24(r, cols) + c; }
struct DepthF32 { const float* p; __device__ float operator()(int32_t i) const { return p[i]; } };
struct DepthU16mm {   // conversions/image_conversions_thrust.cu:39-45 fused into the read
  const uint16_t* p; __device__ float operator()(int32_t i) const { return (float)p[i] * (1.0f / 1000.0f); } };

// returns 1 = value, 0 = no sample (outside the image), -1 = a depth tap is invalid (<= 0)
template <typename Img>
__device__ inline int interp_depth(const Img& img, int rows, int cols, float u, float v, int nearest, float* out) {
  if (nearest) {
    const int c = (int)floorf(u), r = (int)floorf(v);
    if (c < 0 || r < 0 || c >= cols || r >= rows) return 0;
    const float d = img(pix(r, c, cols));
    if (!(d > 0.0f)) return -1;
    *out = d; return 1;
  }
  const float uc = u - 0.5f, vc = v - 0.5f;
  const float fx = floorf(uc), fy = floorf(vc);
  const int x0 = (int)fx, y0 = (int)fy;
  if (x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1) return 0;
  const float ax = uc - fx, ay = vc - fy;
  const int32_t i00 = pix(y0, x0, cols);
  const float f00 = img(i00), f10 = img(i00 + 1), f01 = img(i00 + cols), f11 = img(i00 + cols + 1);
  __builtin_amdgcn_sched_barrier(0);        // both rows' loads in flight before the first tap is looked at (else: two serial round trips)
  if (!(f00 > 0.0f) || !(f10 > 0.0f) || !(f01 > 0.0f) || !(f11 > 0.0f)) return -1;
  const float top = (1.0f - ax) * f00 + ax * f10;
  const float bot = (1.0f - ax) * f01 + ax * f11;
  *out = (1.0f - ay) * top + ay * bot;
  return 1;
}

// [U] ProjectiveOccupancyIntegrator: log-odds update of a voxel at depth `vd` along the ray whose surface was measured at `ds`
constexpr float OCC_LOG_ODDS_CLAMP = 10.0f;
__device__ inline float occupancy_update(const Frame& f, float cur, float ds, float vd) {
  float upd = f.lo_unobserved;
  if (vd < ds - f.occ_half_width) upd = f.lo_free;
  else if (vd <= ds + f.occ_half_width) upd = f.lo_occupied;
  float v = cur + upd;
  if (v > OCC_LOG_ODDS_CLAMP) v = OCC_LOG_ODDS_CLAMP;
  if (v < -OCC_LOG_ODDS_CLAMP) v = -OCC_LOG_ODDS_CLAMP;
  return v;
}

// WeightingFunctionType (mapper_initialization.cpp:31-42).  constant = 1 and inverse-square = 1 / d^2 are unambiguous; the other
// four formulas are [U]/[D] and come in two sets (nvbx_mapper_params::tsdf_weighting_variant; oracle weight_fn, same lines):
//   set A: dropoff = linear ramp 1 -> 0 between the surface and -trunc behind it; tsdf-distance penalty = trunc / sdf for voxels
//          more than trunc in front of the surface; linear-with-max = min(1, 1 / d)
//   set B: dropoff starts one voxel behind the surface, (trunc + sdf) / (trunc - voxel) (the voxblox form); tsdf-distance penalty =
//          quadratic ramp ((trunc + sdf) / trunc)^2 behind the surface; linear-with-max = max(0.01, 1 - d / max_integration_distance)
__device__ inline float weight_fn(int mode, int variant, float d_meas, float d_vox, float trunc, float voxel_size, float max_dist) {
  float w = 1.0f;
  if (mode == 2 || mode == 3 || mode == 4) {
    w = 1.0f / (d_meas * d_meas);
  } else if (mode == 5) {
    if (variant == 0) { w = 1.0f / d_meas; if (w > 1.0f) w = 1.0f; }
    else { w = 1.0f - d_meas / max_dist; if (w < 0.01f) w = 0.01f; }
  }
  const float sdf = d_meas - d_vox;
  if (mode == 1 || mode == 3) {
    if (variant == 0) {
      if (sdf < 0.0f) { float g = (trunc + sdf) / trunc; if (g < 0.0f) g = 0.0f; w = w * g; }
    } else if (sdf < -voxel_size) {
      float g = 0.0f;
      if (trunc > voxel_size) { g = (trunc + sdf) / (trunc - voxel_size); if (g < 0.0f) g = 0.0f; }
      w = w * g;
    }
  } else if (mode == 4) {
    if (variant == 0) { if (sdf > trunc) w = w * (trunc / sdf); }
    else if (sdf < 0.0f) { float g = (trunc + sdf) / trunc; if (g < 0.0f) g = 0.0f; w = w * (g * g); }
  }
  return w;
}
// [U] UpdateTsdfVoxelFunctor: fuses the measurement into *v; false if the measurement does not touch the voxel.
// Switches: skip_at_neg_trunc (the voxel exactly at sdf == -trunc), clamp_before_blend (max_weight clamp order).
__device__ inline bool tsdf_fuse(const Frame& f, float2* v, float ds, float vd) {
  const float2 cur = *v;
  const float sdf = ds - vd;
  if (f.skip_at_neg_trunc ? (sdf <= -f.trunc) : (sdf < -f.trunc)) return false;
  const float wm = weight_fn(f.weighting_mode, f.weighting_variant, ds, vd, f.trunc, f.voxel_size, f.max_dist);
  const float wsum = wm + cur.y;
  if (!(wsum > 0.0f)) return false;
  float fused, wnew = fminf(wsum, f.max_weight);
  if (!f.clamp_before_blend) fused = NVBX_DIV(sdf * wm + cur.x * cur.y, wsum);
  else { float wp = wnew - wm; if (wp < 0.0f) wp = 0.0f; fused = NVBX_DIV(sdf * wm + cur.x * wp, wm + wp); }
  fused = __builtin_amdgcn_fmed3f(fused, -f.trunc, f.trunc);     // = (fused > 0 ? min(trunc, fused) : max(-trunc, fused)), one instruction (a NaN comes out as -trunc either way)
  *v = make_float2(fused, wnew);
  return true;
}

// The same for the configuration almost every caller runs -- constant weighting, clamp after the blend (frame_is_plain) -- with the
// mode switches folded at compile time: identical arithmetic (w = 1: sdf * 1 is sdf), none of the uniform branching.
__device__ inline bool tsdf_fuse_plain(const Frame& f, float2* v, float ds, float vd) {
  const float2 cur = *v;
  const float sdf = ds - vd;
  if (f.skip_at_neg_trunc ? (sdf <= -f.trunc) : (sdf < -f.trunc)) return false;
  const float wsum = 1.0f + cur.y;
  if (!(wsum > 0.0f)) return false;
  float fused = NVBX_DIV(sdf + cur.x * cur.y, wsum);
  fused = __builtin_amdgcn_fmed3f(fused, -f.trunc, f.trunc);     // = (fused > 0 ? min(trunc, fused) : max(-trunc, fused)), one instruction (a NaN comes out as -trunc either way)
  *v = make_float2(fused, fminf(wsum, f.max_weight));
  return true;
}
__host__ __device__ inline bool frame_is_plain(const Frame& f) {
  return !f.occupancy && f.weighting_mode == 0 && !f.clamp_before_blend && !(f.invalid_decay >= 0.0f);
}

// [U] height of the ground plane n . p + d = 0 (nz > 0) at (x, y), fixed evaluation order (the oracle's esdf_plane_height, same lines)
__host__ __device__ inline float esdf_plane_height(const float* pl, float x, float y) {
  float t = pl[0] * x;
  t = t + pl[1] * y;
  t = t + pl[3];
  return -(t / pl[2]);
}
// ESDF packed voxel: {f32 squared_distance_vox, u32 meta}; meta = dx | dy<<8 | dz<<16 (int8 each) | observed<<24 | inside<<25 | site<<26
__device__ __host__ inline uint32_t esdf_meta(int dx, int dy, int dz, int observed, int inside, int site) {
  return ((uint32_t)(uint8_t)(int8_t)dx) | (((uint32_t)(uint8_t)(int8_t)dy) << 8) | (((uint32_t)(uint8_t)(int8_t)dz) << 16) |
         ((uint32_t)(observed != 0) << 24) | ((uint32_t)(inside != 0) << 25) | ((uint32_t)(site != 0) << 26);
}
constexpr uint32_t ESDF_OBSERVED = 1u << 24, ESDF_INSIDE = 1u << 25, ESDF_SITE = 1u << 26, ESDF_FLAG_MASK = 7u << 24;
#endif  // __HIPCC__

}  // namespace nvbx
