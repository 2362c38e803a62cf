// nvblox/integrators/esdf_slicer.h -- EsdfSlicer as called at nvblox_node.cpp:135-137,836-844,149-150,917-919.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nvblox/map/layer.h"
#include "nvblox/sensors/image.h"
#include "nvblox_hip.h"

namespace nvblox {

class EsdfSlicer {
 public:
  EsdfSlicer() = default;
  // AABB of the allocated ESDF blocks -> dense row-major image (row = y, col = x), signed metres or unknown_value
  void sliceLayerToDistanceImage(const EsdfLayer& layer, float slice_height, float unknown_value, AxisAlignedBoundingBox* aabb, Image<float>* image) const {
    (void)slice_height;   // the 2-D ESDF holds exactly one plane (esdf_slice_height of the mapper's parameters)
    int32_t rows = 0, cols = 0; float bb[6] = {0, 0, 0, 0, 0, 0};
    last_ = layer.c_handle();
    checkNvbx(nvbx_esdf_slice_size(layer.c_handle(), &rows, &cols, bb), "nvbx_esdf_slice_size");
    image->resize(rows, cols);
    if (rows > 0 && cols > 0)
      checkNvbx(nvbx_esdf_slice_to_image(layer.c_handle(), unknown_value, image->dataPtr(), (int64_t)rows * cols, &rows, &cols, bb), "nvbx_esdf_slice_to_image");
    if (aabb) *aabb = AxisAlignedBoundingBox({bb[0], bb[1], bb[2]}, {bb[3], bb[4], bb[5]});
  }
  // nvblox_node.cpp:836-840: one costmap from two mappers (static + dynamic): union AABB, min of the observed distances
  void sliceLayersToCombinedDistanceImage(const EsdfLayer& layer_1, const EsdfLayer& layer_2, float slice_height_1, float slice_height_2,
                                          float unknown_value, AxisAlignedBoundingBox* aabb, Image<float>* image) const {
    (void)slice_height_1; (void)slice_height_2;     // each mapper's own esdf_slice_height
    int32_t rows = 0, cols = 0; float bb[6] = {0, 0, 0, 0, 0, 0};
    last_ = layer_1.c_handle();
    checkNvbx(nvbx_esdf_slice_combined_size(layer_1.c_handle(), layer_2.c_handle(), &rows, &cols, bb), "nvbx_esdf_slice_combined_size");
    image->resize(rows, cols);
    if (rows > 0 && cols > 0)
      checkNvbx(nvbx_esdf_slice_combined_to_image(layer_1.c_handle(), layer_2.c_handle(), unknown_value, image->dataPtr(), (int64_t)rows * cols, &rows, &cols, bb),
                "nvbx_esdf_slice_combined_to_image");
    if (aabb) *aabb = AxisAlignedBoundingBox({bb[0], bb[1], bb[2]}, {bb[3], bb[4], bb[5]});
  }
  // nvblox_node.cpp:917-919: int8 occupancy written to HOST memory (the message buffer); 100 occupied, 0 free, -1 unknown.
  // The conversion runs on the GPU (nvbx_occupancy_grid_from_slice); only the int8 result crosses to the host.
  void occupancyGridFromSliceImage(const Image<float>& slice_image, int8_t* occupancy_grid_host, float unknown_value) const {
    const int64_t n = (int64_t)slice_image.numel();
    if (n == 0) return;
    if (!last_) { std::fprintf(stderr, "[nvblox_hip] occupancyGridFromSliceImage needs a slice produced by this EsdfSlicer\n"); std::abort(); }
    int8_t* dev = nullptr;
    (void)hipMalloc((void**)&dev, (size_t)n);
    checkNvbx(nvbx_occupancy_grid_from_slice(last_, slice_image.dataConstPtr(), slice_image.rows(), slice_image.cols(), unknown_value, dev), "nvbx_occupancy_grid_from_slice");
    checkNvbx(nvbx_synchronize(last_), "nvbx_synchronize");
    (void)hipMemcpy(occupancy_grid_host, dev, (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
  }
 private:
  mutable nvbx_mapper* last_ = nullptr;
};

}  // namespace nvblox
