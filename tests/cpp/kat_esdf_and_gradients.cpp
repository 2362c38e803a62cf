// kat_esdf_and_gradients.cpp -- the reference's one golden test of this path, nvblox_ros/test/unit_tests/
// test_esdf_and_gradient_conversions.cpp (FloatGrid :36-83, EsdfValues :110-157), written against the nvblox:: façade +
// libnvblox_hip.so.  No ROS here: `Float32MultiArray` below is the three-dimension layout + data vector of
// std_msgs/Float32MultiArray, and esdfInAabbToMultiArrayMsg is EsdfAndGradientsConverter's method (esdf_and_gradients_conversions.cu:
// 88-125) with the layer -> grid step served by nvbx_esdf_dense_grid.  The reference builds a free-standing EsdfLayer; layers
// here are views of a Mapper, so the layer under test is mapper.esdf_layer().
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>
#include "nvblox/nvblox.h"

using namespace nvblox;

struct MultiArrayDimension { std::string label; uint32_t size = 0, stride = 0; };
struct Float32MultiArray { struct { std::vector<MultiArrayDimension> dim; } layout; std::vector<float> data; };

static int failures = 0;
#define EXPECT_NEAR(a, b, eps) do { const double a_ = (a), b_ = (b); if (!(std::fabs(a_ - b_) <= (eps))) { if (failures++ < 10) std::fprintf(stderr, "%s:%d: |%g - %g| > %g\n", __FILE__, __LINE__, a_, b_, (double)(eps)); } } while (0)

static float getTestValue(const Index3D& idx) {
  constexpr int kMaxValue = 1000;
  return static_cast<float>(Index3DHash()(idx) % kMaxValue);
}

static void setLayout(Float32MultiArray* msg, const Index3D& size_in_voxels) {
  msg->layout.dim.resize(3);
  msg->layout.dim[0].label = "x"; msg->layout.dim[0].size = size_in_voxels.x();
  msg->layout.dim[0].stride = size_in_voxels.x() * size_in_voxels.y() * size_in_voxels.z();
  msg->layout.dim[1].label = "y"; msg->layout.dim[1].size = size_in_voxels.y();
  msg->layout.dim[1].stride = size_in_voxels.y() * size_in_voxels.z();
  msg->layout.dim[2].label = "z"; msg->layout.dim[2].size = size_in_voxels.z();
  msg->layout.dim[2].stride = size_in_voxels.z();
}

// EsdfAndGradientsConverter::esdfInAabbToMultiArrayMsg
struct EsdfAndGradientsConverter {
  Unified3DGrid<float> gpu_grid_{MemoryType::kDevice}, cpu_grid_{MemoryType::kHost};
  Float32MultiArray esdfInAabbToMultiArrayMsg(const EsdfLayer& esdf_layer, const AxisAlignedBoundingBox& aabb, const float default_value,
                                              const CudaStream& cuda_stream) {
    SignedDistanceConversion conversion_op{esdf_layer.voxel_size(), default_value};
    voxelLayerToDenseVoxelGridInAABBAsync(esdf_layer, aabb, default_value, conversion_op, &gpu_grid_, cuda_stream);
    cpu_grid_.copyFromAsync(gpu_grid_, cuda_stream);
    Float32MultiArray array_msg;
    setLayout(&array_msg, gpu_grid_.aabb_size());
    array_msg.data = cpu_grid_.data().toVectorAsync(cuda_stream);
    cuda_stream.synchronize();
    return array_msg;
  }
};

static void testFloatGrid() {
  constexpr int kGridSize = 2;
  Unified3DGrid<float> grid(MemoryType::kUnified);
  grid.setAABB(Index3D(0, 0, 0), Index3D(kGridSize, kGridSize, kGridSize));
  for (int z = 0; z < kGridSize; z++) for (int y = 0; y < kGridSize; y++) for (int x = 0; x < kGridSize; x++) grid(Index3D(x, y, z)) = getTestValue(Index3D(x, y, z));
  Float32MultiArray array_msg;
  setLayout(&array_msg, Index3D(kGridSize, kGridSize, kGridSize));
  array_msg.data = grid.data().toVectorAsync(CudaStreamOwning());
  auto toMsgLinearIdx = [&array_msg](const Index3D& idx) -> int {
    return idx.x() * array_msg.layout.dim[1].stride + idx.y() * array_msg.layout.dim[2].stride + idx.z();
  };
  for (int z = 0; z < kGridSize; z++) for (int y = 0; y < kGridSize; y++) for (int x = 0; x < kGridSize; x++) {
    const Index3D idx(x, y, z);
    EXPECT_NEAR(array_msg.data[toMsgLinearIdx(idx)], grid(idx), 1e-6);
  }
}

static float getValueFromMessage(const Index3D& idx, const Float32MultiArray& msg) {
  const int stride_y = msg.layout.dim[1].stride, stride_z = msg.layout.dim[2].stride;
  return msg.data[idx.z() + idx.y() * stride_z + idx.x() * stride_y];
}

static void testEsdfValues() {
  constexpr float kVoxelSize = 0.05f;
  Mapper mapper(kVoxelSize, MemoryType::kUnified);
  EsdfLayer& esdf_layer = mapper.esdf_layer();
  auto block_ptr = esdf_layer.allocateBlockAtIndex(Index3D(0, 0, 0));
  if (!block_ptr) { failures++; return; }
  callFunctionOnAllVoxels<EsdfVoxel>(&esdf_layer, [](const Index3D&, const Index3D& voxel_index, EsdfVoxel* voxel) {
    voxel->squared_distance_vox = getTestValue(voxel_index);
    voxel->observed = true;
  });
  const auto aabb = getAABBOfAllocatedBlocks(esdf_layer);
  EsdfAndGradientsConverter esdf_and_gradients_converter;
  CudaStreamOwning cuda_stream;
  constexpr float default_value = -1000;
  const Float32MultiArray array_msg = esdf_and_gradients_converter.esdfInAabbToMultiArrayMsg(esdf_layer, aabb, default_value, cuda_stream);
  cuda_stream.synchronize();
  if (array_msg.layout.dim[0].label != "x" || array_msg.layout.dim[1].label != "y" || array_msg.layout.dim[2].label != "z") failures++;
  auto is_inside_block = [](const Index3D& idx) { return idx.x() < 8 && idx.y() < 8 && idx.z() < 8; };
  const Index3D aabb_min_vox((int)(aabb.min().x() / kVoxelSize), (int)(aabb.min().y() / kVoxelSize), (int)(aabb.min().z() / kVoxelSize));
  const Index3D aabb_max_vox((int)(aabb.max().x() / kVoxelSize), (int)(aabb.max().y() / kVoxelSize), (int)(aabb.max().z() / kVoxelSize));
  int checked = 0;
  for (int x = aabb_min_vox.x(); x <= aabb_max_vox.x(); x++) for (int y = aabb_min_vox.y(); y <= aabb_max_vox.y(); y++) for (int z = aabb_min_vox.z(); z <= aabb_max_vox.z(); z++) {
    const Index3D global_voxel_idx(x, y, z);
    const float msg_value = getValueFromMessage(global_voxel_idx, array_msg);
    if (is_inside_block(global_voxel_idx)) EXPECT_NEAR(msg_value, kVoxelSize * std::sqrt(getTestValue(global_voxel_idx)), 1e-6);
    else EXPECT_NEAR(msg_value, default_value, 1e-6);
    checked++;
  }
  if (checked != 9 * 9 * 9) { std::fprintf(stderr, "walked %d voxels, expected 729\n", checked); failures++; }
}

int main() {
  testFloatGrid();
  testEsdfValues();
  std::printf("{\"failures\": %d}\n", failures);
  return failures ? 1 : 0;
}
