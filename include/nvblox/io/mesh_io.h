// nvblox/io/mesh_io.h -- io::outputColorMeshLayerToPly as called by the save_ply service (nvblox_node.cpp:1609-1613):
// ASCII PLY of the mesh last serialized by the mapper (updateColorMesh(UpdateFullLayer::kYes) + serializeSelectedLayers first).
#pragma once
#include <cstdio>
#include <string>
#include "nvblox/mesh/mesh.h"

namespace nvblox {
namespace io {

inline bool outputColorMeshLayerToPly(const SerializedColorMeshLayer& mesh, const std::string& filename) {
  FILE* f = std::fopen(filename.c_str(), "w");
  if (!f) return false;
  const size_t nv = mesh.vertices.size(), nt = mesh.triangle_indices.size() / 3;
  std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
                  "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                  "element face %zu\nproperty list uchar int vertex_indices\nend_header\n", nv, nt);
  for (size_t i = 0; i < nv; i++) {
    const Vector3f& p = mesh.vertices[i]; const Vector3f& n = mesh.vertex_normals[i]; const Color& c = mesh.vertex_appearances[i];
    std::fprintf(f, "%.6f %.6f %.6f %.5f %.5f %.5f %d %d %d\n", p.x(), p.y(), p.z(), n.x(), n.y(), n.z(), (int)c.r, (int)c.g, (int)c.b);
  }
  for (size_t b = 0; b + 1 < mesh.triangle_index_block_offsets.size(); b++) {     // block-local indices -> global
    const int vo = mesh.vertex_block_offsets[b];
    for (int t = mesh.triangle_index_block_offsets[b]; t + 2 < mesh.triangle_index_block_offsets[b + 1]; t += 3)
      std::fprintf(f, "3 %d %d %d\n", vo + mesh.triangle_indices[t], vo + mesh.triangle_indices[t + 1], vo + mesh.triangle_indices[t + 2]);
  }
  return std::fclose(f) == 0;
}

}  // namespace io
}  // namespace nvblox
