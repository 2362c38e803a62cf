// nvblox/serialization/layer_serializer_gpu.h -- include path used by layer_publishing.hpp:30-32.  SerializedLayer<VoxelType>,
// SerializedTsdfLayer / SerializedColorLayer and SerializedColorMeshLayer are defined with the Mapper that fills them.
#pragma once
#include "nvblox/mapper/mapper.h"
#include "nvblox/mesh/mesh.h"
