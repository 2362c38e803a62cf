// nvblox/datasets/data_loader_interface.h -- what nvblox_ros's RosDataLoader derives from and FuserNode branches on
// (nvblox_ros/include/nvblox_ros/rosbag_data_loader.hpp:27-29,77,142-170; src/lib/rosbag_data_loader.cpp:133,179-269;
// src/lib/fuser_node.cpp:216-224).  [U] shape of the core's datasets::RgbdDataLoaderInterface restated from those call sites.
#pragma once
#include "nvblox/core/types.h"
#include "nvblox/sensors/camera.h"
#include "nvblox/sensors/image.h"

namespace nvblox {
namespace datasets {

enum class DataLoadResult { kSuccess, kBadFrame, kNoMoreData };

class RgbdDataLoaderInterface {
 public:
  RgbdDataLoaderInterface() = default;
  virtual ~RgbdDataLoaderInterface() = default;
  // rosbag_data_loader.hpp:131: "The ROSbag data loader does not provide frame timestamps"
  virtual bool provides_frame_timestamps() const { return false; }
  // The general form (depth and colour cameras may differ): rosbag_data_loader.hpp:160-170.  The trailing outputs are the frame time,
  // and -- for loaders that know it -- the pose / time of a second stamp; loaders that have nothing to say leave them untouched.
  virtual DataLoadResult loadNext(DepthImage* depth_frame_ptr, Transform* T_L_D_ptr, Camera* depth_camera_ptr, ColorImage* color_frame_ptr,
                                  Transform* T_L_C_ptr, Camera* color_camera_ptr, Time* depth_time_ptr, Transform* T_L_C_2_ptr, Time* color_time_ptr) = 0;
  // depth and colour share the camera (rosbag_data_loader.hpp:142-146)
  DataLoadResult loadNext(DepthImage* depth_frame_ptr, Transform* T_L_C_ptr, Camera* camera_ptr, ColorImage* color_frame_ptr = nullptr) {
    Transform T_L_color; Camera color_camera;
    ColorImage scratch(MemoryType::kDevice);
    return loadNext(depth_frame_ptr, T_L_C_ptr, camera_ptr, color_frame_ptr ? color_frame_ptr : &scratch, &T_L_color, &color_camera, nullptr, nullptr, nullptr);
  }
  bool setup_success() const { return setup_success_; }

 protected:
  bool setup_success_ = true;
};

}  // namespace datasets
}  // namespace nvblox
