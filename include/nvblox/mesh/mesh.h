// nvblox/mesh/mesh.h -- SerializedColorMeshLayer with the accessors conversions/mesh_conversions.cpp:62-104 uses:
// block_indices, vertices, vertex_appearances, getNumVerticesInBlock, getNumTriangleIndicesInBlock, getVertex,
// getAppearance, getTriangleIndex, and the per-block iterators of the marker path (:149-155: triangleBlockItr(i) .. triangleBlockItr(i + 1)
// walks block i's block-local vertex indices).  Filled from the device mesh arena by Mapper::serializeSelectedLayers.
#pragma once
#include <vector>
#include "nvblox/core/types.h"

namespace nvblox {

struct SerializedColorMeshLayer {
  std::vector<Index3D> block_indices;
  std::vector<Vector3f> vertices;
  std::vector<Vector3f> vertex_normals;
  std::vector<Color> vertex_appearances;
  std::vector<int32_t> triangle_indices;          // 3 per triangle, local to the block
  std::vector<int32_t> vertex_block_offsets;      // n_blocks + 1
  std::vector<int32_t> triangle_index_block_offsets;   // n_blocks + 1 (in indices, not triangles)
  size_t getNumVerticesInBlock(size_t i) const { return (size_t)(vertex_block_offsets[i + 1] - vertex_block_offsets[i]); }
  size_t getNumTriangleIndicesInBlock(size_t i) const { return (size_t)(triangle_index_block_offsets[i + 1] - triangle_index_block_offsets[i]); }
  const Vector3f& getVertex(size_t i_block, size_t i_vert) const { return vertices[(size_t)vertex_block_offsets[i_block] + i_vert]; }
  const Vector3f& getNormal(size_t i_block, size_t i_vert) const { return vertex_normals[(size_t)vertex_block_offsets[i_block] + i_vert]; }
  const Color& getAppearance(size_t i_block, size_t i_vert) const { return vertex_appearances[(size_t)vertex_block_offsets[i_block] + i_vert]; }
  // iterators over one block's range of the flat vectors; *Itr(n_blocks) is the end of the last block
  std::vector<int32_t>::const_iterator triangleBlockItr(size_t i_block) const { return triangle_indices.begin() + triangle_index_block_offsets[i_block]; }
  std::vector<Vector3f>::const_iterator vertexBlockItr(size_t i_block) const { return vertices.begin() + vertex_block_offsets[i_block]; }
  std::vector<Color>::const_iterator appearanceBlockItr(size_t i_block) const { return vertex_appearances.begin() + vertex_block_offsets[i_block]; }
  int32_t getTriangleIndex(size_t i_block, size_t i_tri) const { return triangle_indices[(size_t)triangle_index_block_offsets[i_block] + i_tri]; }
};

}  // namespace nvblox
