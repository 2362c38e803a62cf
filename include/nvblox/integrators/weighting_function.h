// nvblox/integrators/weighting_function.h -- enum named at nvblox_ros/src/lib/mapper_initialization.cpp:31-42.
#pragma once
#include "nvblox_hip.h"
namespace nvblox {
enum class WeightingFunctionType {
  kConstantWeight = NVBX_WEIGHT_CONSTANT,
  kConstantDropoffWeight = NVBX_WEIGHT_CONSTANT_DROPOFF,
  kInverseSquareWeight = NVBX_WEIGHT_INVERSE_SQUARE,
  kInverseSquareDropoffWeight = NVBX_WEIGHT_INVERSE_SQUARE_DROPOFF,
  kInverseSquareTsdfDistancePenalty = NVBX_WEIGHT_INVERSE_SQUARE_TSDF_DISTANCE_PENALTY,
  kLinearWithMax = NVBX_WEIGHT_LINEAR_WITH_MAX
};
}  // namespace nvblox
