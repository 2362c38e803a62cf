// nvblox/map/internal/cuda/impl/layer_to_3d_grid_impl.cuh -- include path of conversions/esdf_and_gradients_conversions.cu:19-23
// (the template implementation header of layer_to_3d_grid.cuh in the reference; here the implementation is in the library).
#pragma once
#include "nvblox/map/internal/cuda/layer_to_3d_grid.cuh"
