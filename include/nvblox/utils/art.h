// nvblox/utils/art.h -- nvblox::art::PrintNvbloxFrog(), logged once by the fuser node when a dataset has been fused
// (fuser_node.cpp:192).  The reference prints its mascot; this build says what ran instead.
#pragma once
#include <string>

namespace nvblox {
namespace art {

inline std::string PrintNvbloxFrog() {
  return "\n"
         "   nvblox  ::  libnvblox_hip (MI355X / gfx950)\n"
         "   [ TSDF | colour | ESDF | mesh ]  voxel-block hash in HBM, wave64 HIP kernels\n";
}

}  // namespace art
}  // namespace nvblox
