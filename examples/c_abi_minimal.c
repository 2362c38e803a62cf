/* c_abi_minimal.c -- the drop-in boundary from plain C (gcc, no C++, no HIP compiler): one depth + colour frame, an ESDF update,
 * the costmap slice and the mesh sizes through include/nvblox_hip.h only.  Device buffers come from the HIP runtime's C API.
 *   make -C tests/cpp c_abi_minimal && tests/cpp/c_abi_minimal                                                        */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "nvblox_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nvbx_last_error()); return 1; } } while (0)

int main(void) {
  enum { ROWS = 120, COLS = 160 };
  nvbx_mapper_params p; nvbx_default_params(&p);                 /* fuser.yaml values */
  nvbx_mapper* m = NULL;
  CHECK(nvbx_mapper_create(0, NULL, &p, 1 << 12, &m));           /* NULL stream: the library creates its own */
  /* a wall 2 m in front of the camera; camera z = layer x, 1 m above the floor */
  float* depth = (float*)malloc(sizeof(float) * ROWS * COLS);
  unsigned char* rgb = (unsigned char*)malloc(3 * ROWS * COLS);
  for (int i = 0; i < ROWS * COLS; i++) { depth[i] = 2.0f; rgb[3 * i] = 200; rgb[3 * i + 1] = 120; rgb[3 * i + 2] = 40; }
  float* d_depth; unsigned char* d_rgb;
  if (hipMalloc((void**)&d_depth, sizeof(float) * ROWS * COLS) != hipSuccess || hipMalloc((void**)&d_rgb, 3 * ROWS * COLS) != hipSuccess) return 1;
  hipMemcpy(d_depth, depth, sizeof(float) * ROWS * COLS, hipMemcpyHostToDevice);
  hipMemcpy(d_rgb, rgb, 3 * ROWS * COLS, hipMemcpyHostToDevice);
  const nvbx_camera cam = {80.f, 80.f, 79.5f, 59.5f, COLS, ROWS};
  const float T_L_C[16] = {0, 0, 1, 0,  -1, 0, 0, 0,  0, -1, 0, 1.0f,  0, 0, 0, 1};
  CHECK(nvbx_integrate_depth(m, d_depth, ROWS, COLS, T_L_C, &cam));
  CHECK(nvbx_integrate_color(m, d_rgb, ROWS, COLS, T_L_C, &cam));
  CHECK(nvbx_update_esdf(m));
  CHECK(nvbx_update_color_mesh(m, 0));
  int32_t rows = 0, cols = 0; float aabb[6];
  CHECK(nvbx_esdf_slice_size(m, &rows, &cols, aabb));
  float* slice = (float*)malloc(sizeof(float) * (size_t)rows * cols);
  CHECK(nvbx_esdf_slice_to_host(m, 1000.0f, slice, (int64_t)rows * cols, &rows, &cols, aabb));
  int known = 0; float dmin = 1e9f;
  for (int i = 0; i < rows * cols; i++) if (slice[i] < 999.0f) { known++; if (slice[i] < dmin) dmin = slice[i]; }
  int64_t nb = 0, nv = 0, nt = 0;
  CHECK(nvbx_mesh_sizes(m, &nb, &nv, &nt));
  nvbx_counters c;
  CHECK(nvbx_get_counters(m, &c));
  printf("{\"tsdf_blocks\": %lld, \"slice_rows\": %d, \"slice_cols\": %d, \"slice_known\": %d, \"slice_min_m\": %.4f, \"mesh_blocks\": %lld, \"mesh_vertices\": %lld, \"mesh_triangles\": %lld}\n",
         (long long)nvbx_num_blocks(m, NVBX_LAYER_TSDF), rows, cols, known, dmin, (long long)nb, (long long)nv, (long long)nt);
  CHECK(nvbx_mapper_destroy(m));
  hipFree(d_depth); hipFree(d_rgb); free(depth); free(rgb); free(slice);
  return (known > 100 && nv > 100) ? 0 : 2;
}
