/* nvblox_hip_device.h -- device-side read access to a libnvblox_hip map from the caller's OWN HIP kernels.
 *
 * Replaces, for this library, what nvblox_ros gets from nvblox/gpu_hash/internal/cuda/gpu_indexing.cuh and
 * GPULayerView<Block> (call sites: nvblox_ros/src/lib/conversions/esdf_slice_conversions.cu:18,
 * esdf_and_gradients_conversions.cu:19-23,88-125): a kernel that is not part of the library looks blocks up through
 * the hash and reads voxels in place -- no copy of the layer, no host round trip.
 *
 * Contract: fill an nvbx_device_view with nvbx_get_device_view() (include/nvblox_hip.h) and pass it BY VALUE to a kernel
 * launched on the mapper's stream (or ordered behind it with an event).  The view's pointers are valid until the next
 * call that can allocate blocks (the pools and the table grow on demand, nvbx_mapper_set_max_capacity in nvblox_hip.h) or
 * that decays the map (nvbx_decay_tsdf / nvbx_decay_occupancy build the table of the surviving blocks in a second buffer
 * and the two change places): fetch the view again after an integrate / upload / decay call, it costs nothing.  What a kernel reads through it is whatever the mapper calls enqueued before the kernel have
 * produced.  Read-only: writing through the view breaks the library's invariants.
 * Layouts: TSDF / colour voxel v = z + 8y + 64x in the block (the reference's order); ESDF voxel
 * v = x + 8y + 64z, packed {f32 squared_distance_vox, u32 meta}. */
#ifndef NVBLOX_HIP_DEVICE_H_
#define NVBLOX_HIP_DEVICE_H_
#include <stdint.h>
#include "nvblox_hip.h"

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

#define NVBX_DEV_SLOT_NONE 0xFFFFFFFEu     /* values >= this are "no block" */

/* slot of block (x, y, z) if it carries `layer` (NVBX_LAYER_*), else NVBX_DEV_SLOT_NONE.  Probe start = the reference's
 * Index3DHash (nvblox_rviz_plugin/.../nvblox_hash_utils.h:43-48) scattered by a Fibonacci multiply; linear probing. */
__device__ inline uint32_t nvbx_dev_find_block(const nvbx_device_view& v, int32_t x, int32_t y, int32_t z, uint32_t layer) {
  const unsigned long long B = 1ull << 20;
  const unsigned long long key = (((unsigned long long)(long long)x + B) & 0x1FFFFFull) | ((((unsigned long long)(long long)y + B) & 0x1FFFFFull) << 21) |
                                 ((((unsigned long long)(long long)z + B) & 0x1FFFFFull) << 42);
  uint32_t h = (((uint32_t)x + (uint32_t)y * 17191u + (uint32_t)z * (17191u * 17191u)) * 2654435761u) >> v.table_shift;
  for (uint32_t probe = 0; probe <= v.table_mask; ++probe) {
    const uint4 e = reinterpret_cast<const uint4*>(v.table)[h];          /* {key lo, key hi, slot, view stamp} */
    const unsigned long long k = ((unsigned long long)e.y << 32) | (unsigned long long)e.x;
    if (k == key) return (e.z < NVBX_DEV_SLOT_NONE && (v.slot_flags[e.z] & layer)) ? e.z : NVBX_DEV_SLOT_NONE;
    if (k == ~0ull) return NVBX_DEV_SLOT_NONE;
    h = (h + 1) & v.table_mask;
  }
  return NVBX_DEV_SLOT_NONE;
}
__device__ inline bool nvbx_dev_slot_ok(uint32_t slot) { return slot < NVBX_DEV_SLOT_NONE; }

/* voxel (vx, vy, vz) in 0..7 of a block found above */
__device__ inline nvbx_tsdf_voxel nvbx_dev_tsdf_voxel(const nvbx_device_view& v, uint32_t slot, int vx, int vy, int vz) {
  return reinterpret_cast<const nvbx_tsdf_voxel*>(v.tsdf)[(size_t)slot * 512 + vz + 8 * vy + 64 * vx];
}
__device__ inline nvbx_color_voxel nvbx_dev_color_voxel(const nvbx_device_view& v, uint32_t slot, int vx, int vy, int vz) {
  return reinterpret_cast<const nvbx_color_voxel*>(v.color)[(size_t)slot * 512 + vz + 8 * vy + 64 * vx];
}
/* ESDF voxel unpacked to the reference's fields (esdf_and_gradients_conversions.cu:28-48 reads squared_distance_vox, observed, is_inside) */
__device__ inline nvbx_esdf_voxel nvbx_dev_esdf_voxel(const nvbx_device_view& v, uint32_t slot, int vx, int vy, int vz) {
  const uint2 p = reinterpret_cast<const uint2*>(v.esdf)[(size_t)slot * 512 + vx + 8 * vy + 64 * vz];
  nvbx_esdf_voxel o;
  o.squared_distance_vox = __uint_as_float(p.x);
  o.parent_direction[0] = (int8_t)(p.y & 0xFF); o.parent_direction[1] = (int8_t)((p.y >> 8) & 0xFF); o.parent_direction[2] = (int8_t)((p.y >> 16) & 0xFF);
  o.observed = (p.y >> 24) & 1u; o.is_inside = (p.y >> 25) & 1u; o.is_site = (p.y >> 26) & 1u; o.pad = 0;
  return o;
}
/* signed ESDF distance in metres at global voxel index (gx, gy, gz), or `unknown_value` (SignedDistanceFunctor,
 * esdf_and_gradients_conversions.cu:28-48) */
__device__ inline float nvbx_dev_esdf_distance_m(const nvbx_device_view& v, int32_t gx, int32_t gy, int32_t gz, float unknown_value) {
  const uint32_t s = nvbx_dev_find_block(v, gx >> 3, gy >> 3, gz >> 3, NVBX_LAYER_ESDF);
  if (!nvbx_dev_slot_ok(s)) return unknown_value;
  const nvbx_esdf_voxel e = nvbx_dev_esdf_voxel(v, s, gx & 7, gy & 7, gz & 7);
  if (!e.observed) return unknown_value;
  const float d = sqrtf(e.squared_distance_vox) * v.voxel_size;
  return e.is_inside ? -d : d;
}
#endif  /* __HIPCC__ */
#endif  /* NVBLOX_HIP_DEVICE_H_ */
