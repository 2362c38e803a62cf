// nvblox/map/internal/cuda/layer_to_3d_grid.cuh -- include path of conversions/esdf_and_gradients_conversions.cu:19-23.
// voxelLayerToDenseVoxelGridInAABBAsync (the dense ESDF-in-AABB query, esdf_and_gradients_conversions.cu:88-125) is provided by
// the facade over nvbx_esdf_dense_grid (include/nvblox/map/unified_3d_grid.h); the device-side accessors by nvblox_hip_device.h.
#pragma once
#include "nvblox_hip_device.h"
#include "nvblox/map/layer.h"
#include "nvblox/map/unified_3d_grid.h"
