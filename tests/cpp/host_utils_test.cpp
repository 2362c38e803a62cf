// host_utils_test.cpp -- the host-only utilities of the façade that nvblox_ros uses around the hot path (no GPU needed):
// timing::Rates / Delays (nvblox_node.cpp:72-75,469-477,179-180), parameters::ParameterTreeNode + MapperParams::getParameterTree
// (nvblox_node.hpp:556, node_params.cpp:36-43, nvblox_node.cpp:119-124), conversions::saveOccupancyGridAsPng / Yaml (:156-166).
#include <cmath>
#include <cstdio>
#include <string>
#include "nvblox/core/parameter_tree.h"
#include "nvblox/integrators/occupancy_conversions.h"
#include "nvblox/io/layer_io.h"
#include <memory>
#include "nvblox/mapper/mapper_params.h"
#include "nvblox/mesh/mesh.h"
#include "nvblox/utils/art.h"
#include "nvblox/utils/delays.h"
#include "nvblox/utils/rates.h"

using namespace nvblox;
static int failures = 0;
#define CHECK_T(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

int main(int argc, char** argv) {
  const std::string out_dir = argc > 1 ? argv[1] : "/tmp";
  // Rates on a replaced clock (the node passes the ROS clock): 11 ticks 50 ms apart = 20 Hz
  uint64_t fake_now = 1000;
  timing::Rates::setGetTimestampFunctor([&fake_now]() -> uint64_t { return fake_now; });
  CHECK_T(timing::Rates::getMeanRateHz("ros/tick") == 0.0f);
  for (int i = 0; i < 11; i++) { timing::Rates::tick("ros/tick"); fake_now += 50000000ull; }
  CHECK_T(std::fabs(timing::Rates::getMeanRateHz("ros/tick") - 20.0f) < 1e-3f);
  for (int i = 0; i < 300; i++) { timing::Rates::tick("ros/tick"); fake_now += 10000000ull; }     // window slides: 100 Hz now
  CHECK_T(std::fabs(timing::Rates::getMeanRateHz("ros/tick") - 100.0f) < 1e-2f);
  CHECK_T(timing::Rates::Print().find("ros/tick") != std::string::npos);
  // Delays: nanosecond stamps
  timing::Delays::tick("ros/depth_image_callback", Time(1000000000), Time(1030000000));
  timing::Delays::tick("ros/depth_image_callback", Time(2000000000), Time(2010000000));
  CHECK_T(std::fabs(timing::Delays::getMeanDelaySeconds("ros/depth_image_callback") - 0.020) < 1e-9);
  CHECK_T(timing::Delays::Print().find("ros/depth_image_callback") != std::string::npos);
  // parameter tree as the node builds it
  parameters::ParameterTreeNode parameter_tree_{"nvblox_node", {}};
  CHECK_T(parameter_tree_.children().has_value());
  parameter_tree_.children().value().push_back(parameters::ParameterTreeNode("voxel_size", 0.05f));
  parameter_tree_.children().value().push_back(parameters::ParameterTreeNode("global_frame", std::string("odom")));
  parameter_tree_.children().value().push_back(parameters::ParameterTreeNode("use_lidar", false));
  MapperParams mp; mp.esdf_integrator_params.esdf_slice_height = 0.09f;
  parameter_tree_.children().value().push_back(mp.getParameterTree("static_mapper"));
  const std::string txt = parameters::parameterTreeToString(parameter_tree_);
  CHECK_T(txt.find("nvblox_node:\n  voxel_size: 0.05\n  global_frame: odom\n  use_lidar: false\n  static_mapper:\n") == 0);
  CHECK_T(txt.find("      esdf_slice_height: 0.09\n") != std::string::npos);
  CHECK_T(txt.find("      projective_integrator_max_weight: 5\n") != std::string::npos);
  // occupancy grid: 3 rows x 4 cols; row 0 (y = min) occupied, row 1 free, row 2 unknown
  std::vector<int8_t> grid = {100, 100, 100, 100, 0, 0, 0, 0, -1, -1, -1, -1};
  CHECK_T(conversions::saveOccupancyGridAsPng(out_dir + "/nvbx_occ.png", 0.25f, 0.65f, 3, 4, grid));
  CHECK_T(conversions::saveOccupancyGridYaml(out_dir + "/nvbx_occ.yaml", "nvbx_occ.png", 0.05f, -1.2f, 0.4f, 0.25f, 0.65f));
  CHECK_T(!conversions::saveOccupancyGridAsPng(out_dir + "/nvbx_occ_bad.png", 0.25f, 0.65f, 0, 4, grid));
  // a map larger than one stored deflate block (65535 B): 300 x 301, value pattern by position
  std::vector<int8_t> big((size_t)300 * 301);
  for (size_t r = 0; r < 300; r++) for (size_t c = 0; c < 301; c++) big[r * 301 + c] = (int8_t)(((r + 2 * c) % 3 == 0) ? 100 : (((r + 2 * c) % 3 == 1) ? 0 : -1));
  CHECK_T(conversions::saveOccupancyGridAsPng(out_dir + "/nvbx_occ_big.png", 0.25f, 0.65f, 300, 301, big));
  // SerializedColorMeshLayer per-block iteration as the marker path does it (mesh_conversions.cpp:149-155)
  SerializedColorMeshLayer mesh;
  mesh.block_indices = {Index3D(0, 0, 0), Index3D(1, 0, 0)};
  mesh.vertices = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {5, 5, 5}, {6, 5, 5}, {5, 6, 5}, {5, 5, 6}};
  mesh.vertex_normals = mesh.vertices; mesh.vertex_appearances.assign(7, Color(1, 2, 3)); mesh.vertex_appearances[4] = Color(9, 8, 7);
  mesh.vertex_block_offsets = {0, 3, 7}; mesh.triangle_indices = {0, 1, 2, 0, 1, 2, 1, 2, 3}; mesh.triangle_index_block_offsets = {0, 3, 9};
  int walked = 0; float sum_x = 0.f; int reds = 0;
  for (size_t i_block = 0; i_block < mesh.block_indices.size(); i_block++)
    for (auto itr = mesh.triangleBlockItr(i_block); itr != mesh.triangleBlockItr(i_block + 1); ++itr) {
      sum_x += mesh.getVertex(i_block, *itr).x(); reds += mesh.getAppearance(i_block, *itr).r; walked++;
    }
  CHECK_T(walked == 9 && mesh.getNumTriangleIndicesInBlock(1) == 6 && mesh.getNumVerticesInBlock(1) == 4);
  CHECK_T(std::fabs(sum_x - (1.f + (5 + 6 + 5) + (6 + 5 + 5))) < 1e-6f && reds == 7 * 1 + 2 * 9);
  CHECK_T(mesh.vertexBlockItr(1) - mesh.vertexBlockItr(0) == 3 && mesh.appearanceBlockItr(2) == mesh.vertex_appearances.end());
  CHECK_T(art::PrintNvbloxFrog().find("nvblox") != std::string::npos);
  // outputVoxelLayerToPly over a stand-in layer (the façade's views have the same four members, backed by the GPU map)
  struct FakeEsdfLayer {
    float voxel_size() const { return 0.05f; } float block_size() const { return 0.4f; }
    std::vector<Index3D> getAllBlockIndices() const { return {Index3D(1, 0, -1), Index3D(7, 7, 7)}; }
    std::shared_ptr<VoxelBlock<EsdfVoxel>> getBlockAtIndex(const Index3D& b) const {
      if (b.x() != 1) return nullptr;                                   // a listed block that vanished: skipped
      auto blk = std::make_shared<VoxelBlock<EsdfVoxel>>();
      blk->voxels[0][0][0].observed = true; blk->voxels[0][0][0].squared_distance_vox = 16.f;                      // +0.2 m
      blk->voxels[7][3][2].observed = true; blk->voxels[7][3][2].squared_distance_vox = 4.f; blk->voxels[7][3][2].is_inside = true;   // -0.1 m
      blk->voxels[1][1][1].squared_distance_vox = 9.f;                   // not observed: no point
      return blk;
    }
  } fake_layer;
  CHECK_T(io::outputVoxelLayerToPly(fake_layer, out_dir + "/nvbx_esdf.ply"));
  CHECK_T(!io::outputVoxelLayerToPly(fake_layer, out_dir + "/no/such/dir/x.ply"));
  std::printf("{\"failures\": %d}\n", failures);
  return failures ? 1 : 0;
}
