// nvblox/gpu_hash/internal/cuda/gpu_indexing.cuh -- include path of conversions/esdf_slice_conversions.cu:18.  The device-side block
// lookup / voxel accessors the caller's own kernels use (the role of GPULayerView + gpu_indexing.cuh) live in
// include/nvblox_hip_device.h; this forwarder lets the reference's .cu sources keep their #include lines when they are compiled
// with hipcc (INTEGRATION.md shows the kernel-side edits: nvbx_device_view instead of GPULayerView).
#pragma once
#include "nvblox_hip_device.h"
#include "nvblox/map/layer.h"
