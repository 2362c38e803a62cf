/*
 * nvblox_oracle.c -- CPU restatement of the nvblox_core TSDF / Color / ESDF-2D /
 * Mesh hot path.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is linked,
 * imported or executed by the product library (libnvblox_hip.so) or its host
 * code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * PARITY STATUS: *unpinned* for TSDF / colour / ESDF propagation / mesh.  The
 * arithmetic of this path lives in the un-vendored submodule
 * nvidia-isaac/nvblox (branch `public`, commit not recoverable; contemporaneous
 * with isaac_ros_nvblox 4.3.0 -- /root/reference/.gitmodules:1-4,
 * nvblox_ros/package.xml:24).  This file restates the published algorithms of
 * that module and anchors every data-layout / parameter decision on the
 * reference's own call sites.  The single golden test the reference holds for
 * this path (nvblox_ros/test/unit_tests/test_esdf_and_gradient_conversions.cpp:
 * 36-157) is reproduced in tests/test_oracle_kat.py against orc_esdf_dense_grid().
 * Also restated here, equally unpinned ([U] everywhere the core's source would be needed): the LiDAR integrator, occupancy
 * mappers and their decay, the 3-D ESDF, the mask split of the human mapping types, the freespace layer / dynamics detection /
 * connected-component clean-up of the dynamic mapping type, decay and clearing.
 *
 * Anchors inside /root/reference (file:line):
 *   block = 8x8x8 voxels, linear index z + 8*y + 64*x ... nvblox_ros/src/lib/layer_publishing.cpp:335,501
 *   voxel centre = bi*block_size + i*voxel_size + voxel_size/2 ... layer_publishing.cpp:527-529
 *   block hash x + 17191*y + 17191^2*z (uint32) ... nvblox_rviz_plugin/include/nvblox_rviz_plugin/nvblox_hash_utils.h:40-50
 *   TsdfVoxel{distance,weight}, ColorVoxel{color,weight}, EsdfVoxel{squared_distance_vox,observed,is_inside,...}
 *        ... layer_publishing.cpp:62-76,111,179,192; conversions/esdf_and_gradients_conversions.cu:28-48
 *   Camera(fu,fv,cu,cv,w,h) from K[0],K[4],K[2],K[5] ... conversions/image_conversions.cpp:27-32
 *   integrator knobs ... mapper_initialization.cpp:231-466; values nvblox_examples_bringup/config/nvblox/fuser.yaml:24-42
 *   ESDF signed metres = +-sqrt(sq)*voxel_size, default when !observed ... esdf_and_gradients_conversions.cu:33-44
 *   slice image row=y, col=x, origin aabb.min ... conversions/esdf_slice_conversions.cu:60-64; nvblox_msgs/msg/DistanceMapSlice.msg:9-30
 *   dense grid linearisation x*(Ny*Nz)+y*Nz+z ... esdf_and_gradients_conversions.cu:110-119
 *
 * Floating point: compiled with -ffp-contract=off; every expression below is
 * written in the evaluation order that DESIGN.md ("numerical contract") fixes
 * so that the HIP kernels (also contraction-off, IEEE div/sqrt) can be
 * compared bit-for-bit on indices and to 1e-4 on values.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
/* Source SHARED with the product (product headers included by test infrastructure, not the other way round), so that both sides produce identical
 * bits where libm and the device library differ in the last ulp -- and what checks each shared piece WITHOUT sharing anything with it:
 *   csrc/nvbx_lidar_math.h   LiDAR projection (asin / atan2 polynomials)          tests/lidar_independent.py: float64 numpy + libm, 5e5 points (tests/test_independent_checks.py, tests/test_lidar.py)
 *   csrc/nvbx_motion_math.h  pose interpolation of the motion compensation      tests/motion_independent.py: float64 numpy + scipy.Rotation, 2.4e5 points against this file, 3e5 against the kernel (tests/test_lidar.py)
 *   mc_table*.inc            marching-cubes tables (tools/gen_mc_table.py)        brute-force sign topology of all 256 cases, crack census over the 4096 two-cube configurations (tests/test_independent_checks.py)
 * A DEFINITION shared with the product rather than source: the view calculation's crossing parameters in closed form (raycast_blocks below <->
 * csrc/tsdf.hip dda_step) -- checked against float64 segment / grid-plane geometry without any stepping, tests/view_independent.py
 * (tests/test_independent_checks.py::test_view_calculation_against_float64_geometry), and against the textbook accumulated traversal kept below behind
 * orc_set_traversal_accumulate (::test_closed_form_traversal_against_textbook_accumulation: 0 of 111 735 blocks differ on a 200 m scan).
 * The TSDF update rule of this file (tsdf_integrate_block: all six weighting modes, both formula sets, blend, clamps) has a float64 numpy model of its
 * own, tests/tsdf_independent.py (::test_tsdf_update_rule_against_an_independent_float64_model, > 1.5 M voxels within 2e-5).
 * Further rules of this file with a numpy model of their own (tests/test_independent_checks.py; none shares code with this file or the product): occupancy
 * log-odds update and invalid-depth decay (tests/tsdf_independent.py), 2-D ESDF column marking, the colour voxel rule, TSDF decay with its three switches,
 * radius and shape clearing, the freespace state machine + dynamic mask, and the mesh integrator's table-free rules (tests/mesh_independent.py: meshed
 * cubes, welded vertices at the zero crossing in edge order, triangle containment / orientation, vertex colour, normal rule 0).
 * (csrc/nvbx_arith.h is NOT used here: this file keeps `/` and sqrtf; the kernel's shortened sequences are compared with numpy's IEEE results, tests/test_gpu_arith.py.) */
#include "../isaac_ros_nvblox_amd/csrc/nvbx_lidar_math.h"
#include "../isaac_ros_nvblox_amd/csrc/nvbx_motion_math.h"

#define VPS 8
#define NVOX 512

typedef struct { int32_t x, y, z; } Idx3;
typedef struct { float distance, weight; } TsdfVoxel;
typedef struct { uint8_t r, g, b, pad; float weight; } ColorVoxel;
typedef struct { float sq; int32_t parent[3]; uint8_t is_inside, observed, is_site, pad; } EsdfVoxel;
/* FreespaceVoxel (layer_publishing.cpp:129-137,158-165: consecutive_occupancy_duration_ms, is_high_confidence_freespace) */
typedef struct { int64_t last_occupied_timestamp_ms; int64_t consecutive_occupancy_duration_ms; uint8_t is_high_confidence_freespace; uint8_t initialized; uint8_t pad[6]; } FreespaceVoxel;

/* Parameter block: mirrors the reference's MapperParams fields that touch this path
 * (mapper_initialization.cpp:246-380).  Layout shared with tests via ctypes. */
typedef struct {
  float voxel_size;
  float max_integration_distance_m;     /* projective_integrator_max_integration_distance_m */
  float truncation_distance_vox;        /* projective_integrator_truncation_distance_vox */
  float max_weight;                     /* projective_integrator_max_weight */
  int32_t weighting_mode;               /* WeightingFunctionType, mapper_initialization.cpp:31-42 */
  int32_t raycast_subsampling_factor;   /* view calculator */
  float esdf_min_weight;                /* esdf_integrator_min_weight */
  float esdf_max_site_distance_vox;     /* esdf_integrator_max_site_distance_vox */
  float esdf_max_distance_m;            /* esdf_integrator_max_distance_m */
  float esdf_slice_height;              /* esdf_slice_height */
  float esdf_slice_min_height;          /* esdf_slice_min_height */
  float esdf_slice_max_height;          /* esdf_slice_max_height */
  float mesh_min_weight;                /* mesh_integrator_min_weight */
  int32_t mesh_weld_vertices;           /* mesh_integrator_weld_vertices */
  int32_t sphere_tracing_subsampling;   /* projective_color_integrator sphere-tracing ray subsampling (4) */
  int32_t sphere_tracing_max_steps;     /* 100 */
  float sphere_tracing_max_ray_length_m;/* 15 */
  float sphere_tracing_surface_eps_vox; /* 0.1 */
  float tsdf_decay_factor;              /* tsdf_decay_factor */
  float tsdf_decayed_weight_threshold;  /* tsdf_decayed_weight_threshold */
  int32_t esdf_site_rule;               /* 0: inside && |d|<=max_site (default, [U] recall); 1: |d|<=max_site */
  int32_t depth_interp_nearest;         /* 0: bilinear-with-validity (default); 1: nearest */
  float lidar_max_integration_distance_m;  /* lidar_projective_integrator_max_integration_distance_m, mapper_initialization.cpp:271-276 */
  float lidar_linear_interpolation_max_allowable_difference_vox;   /* [U] 2.0 */
  float lidar_nearest_interpolation_max_allowable_dist_to_ray_vox; /* [U] 0.5 */
  int32_t workspace_bounds_type;        /* 0 unbounded, 1 height_bounds, 2 bounding_box (mapper_initialization.cpp:62-80) */
  float workspace_bounds_min_corner_m[3];
  float workspace_bounds_max_corner_m[3];
  int32_t do_depth_preprocessing;       /* do_depth_preprocessing */
  int32_t depth_preprocessing_num_dilations;
  float invalid_depth_decay_factor;     /* projective_tsdf_integrator_invalid_depth_decay_factor; < 0 = off */
  int32_t projective_layer_type;        /* 0 = TSDF, 1 = occupancy */
  float free_region_occupancy_probability, occupied_region_occupancy_probability, unobserved_region_occupancy_probability;
  float occupied_region_half_width_m;
  float free_region_decay_probability, occupied_region_decay_probability;
  int32_t esdf_mode;                    /* 0 = 2-D slice, 1 = 3-D */
  /* freespace integrator (projective_layer_type 2 = TSDF with freespace; mapper_initialization.cpp:430-462, nvblox_dynamics.yaml:12-18) */
  float max_tsdf_distance_for_occupancy_m;
  int32_t max_unobserved_to_keep_consecutive_occupancy_ms, min_duration_since_occupied_for_freespace_ms,
          min_consecutive_occupancy_duration_for_reset_ms, check_neighborhood, initialize_to_high_confidence_freespace;
  /* [U] open choices as switches (include/nvblox_hip.h, same names): every one implemented here and in the HIP kernels */
  int32_t tsdf_weighting_variant, tsdf_skip_at_negative_truncation, tsdf_weight_clamp_before_blend;
  float color_occlusion_threshold_vox;
  int32_t esdf_propagation, mesh_ambiguity_rule, mesh_normal_rule;
  /* decay integrator switches (mapper_initialization.cpp:383-428) -- [U] semantics, same lines in maintenance.hip k_decay */
  int32_t decay_deallocate_decayed_blocks, tsdf_set_free_distance_on_decayed; float tsdf_decayed_free_distance_vox; int32_t occupancy_decay_to_free;
  /* ground-plane-relative 2-D slice (mapper_initialization.cpp:136,257-260) -- [U] semantics, same lines in csrc/nvbx_esdf_mark.h */
  float slice_height_above_plane_m, slice_height_thickness_m; int32_t esdf_use_ground_plane; float esdf_ground_plane[4];
} OrcParams;

enum { W_CONSTANT = 0, W_CONSTANT_DROPOFF = 1, W_INVERSE_SQUARE = 2, W_INVERSE_SQUARE_DROPOFF = 3,
       W_INVERSE_SQUARE_TSDF_DISTANCE_PENALTY = 4, W_LINEAR_WITH_MAX = 5 };

typedef struct MeshBlock {
  int32_t n_vert, n_tri;
  float* vert;    /* n_vert*3 */
  float* nrm;     /* n_vert*3 */
  uint8_t* col;   /* n_vert*4 rgba */
  int32_t* tri;   /* n_tri*3 */
} MeshBlock;

typedef struct Block {
  Idx3 idx;
  uint32_t flags;         /* bit0 tsdf, bit1 color, bit2 esdf, bit3 mesh */
  TsdfVoxel* tsdf;        /* [512] */
  ColorVoxel* color;      /* [512] */
  EsdfVoxel* esdf;        /* [512] */
  FreespaceVoxel* fs;     /* [512] */
  MeshBlock mesh;
  int32_t stamp_view, stamp_esdf, dirty_esdf, dirty_mesh, remark_esdf;
  int32_t stamp_cam;                           /* last CAMERA depth frame that had the block in view (a LiDAR scan does not touch it) */
} Block;

enum { L_TSDF = 1, L_COLOR = 2, L_ESDF = 4, L_MESH = 8, L_FREESPACE = 32 };

typedef struct {
  OrcParams p;
  Block** table; int64_t cap; int64_t count;   /* open addressing, hash = x + 17191 y + 17191^2 z */
  Block** order; int64_t order_cap;            /* insertion order */
  int32_t frame, esdf_epoch;
  int32_t camera_frame;                        /* frame stamp of the last CAMERA depth frame (decayTsdfExcludeLastView<Camera>, nvblox_node.cpp:931-936) */
  int64_t time_ms;                             /* update_time_ms of the next integrateDepth (freespace layer) */
  Idx3* view; int64_t n_view, view_cap;        /* blocks in view of last depth frame */
  Idx3* cview; int64_t n_cview, cview_cap;     /* blocks updated by last colour frame */
  float* synth; int synth_rows, synth_cols;    /* last synthetic depth image (sphere tracing) */
  Idx3* cleared; int64_t n_cleared, cleared_cap; /* Mapper::getClearedBlocks (layer_publishing.cpp:716): projective blocks deallocated since the last take */
} OrcMap;

/* one table per ambiguity rule (OrcParams.mesh_ambiguity_rule; tools/gen_mc_table.py) */
static const int8_t MC_TRI[3][256][16] = {{
#include "mc_table.inc"
}, {
#include "mc_table_r1.inc"
}, {
#include "mc_table_r2.inc"
}};

/* ------------------------------------------------------------------ hashing */
static inline uint32_t idx_hash(Idx3 i) {
  /* nvblox_hash_utils.h:43-48: static_cast<unsigned int>(x + y*sl + z*sl2) with size_t arithmetic */
  const uint64_t sl = 17191ull, sl2 = sl * sl;
  return (uint32_t)((uint64_t)(int64_t)i.x + (uint64_t)(int64_t)i.y * sl + (uint64_t)(int64_t)i.z * sl2);
}
uint32_t orc_index_hash(int32_t x, int32_t y, int32_t z) { Idx3 i = {x, y, z}; return idx_hash(i); }

static Block* map_find(const OrcMap* m, Idx3 i) {
  if (!m->cap) return NULL;
  uint64_t h = idx_hash(i) & (uint64_t)(m->cap - 1);
  for (;;) {
    Block* b = m->table[h];
    if (!b) return NULL;
    if (b->idx.x == i.x && b->idx.y == i.y && b->idx.z == i.z) return b;
    h = (h + 1) & (uint64_t)(m->cap - 1);
  }
}
static void map_put_raw(OrcMap* m, Block* b) {
  uint64_t h = idx_hash(b->idx) & (uint64_t)(m->cap - 1);
  while (m->table[h]) h = (h + 1) & (uint64_t)(m->cap - 1);
  m->table[h] = b;
}
static Block* map_get_or_create(OrcMap* m, Idx3 i) {
  Block* b = map_find(m, i);
  if (b) return b;
  if ((m->count + 1) * 2 > m->cap) {
    int64_t ncap = m->cap ? m->cap * 2 : 1024;
    Block** old = m->table; int64_t ocap = m->cap;
    m->table = (Block**)calloc((size_t)ncap, sizeof(Block*)); m->cap = ncap;
    for (int64_t k = 0; k < ocap; k++) if (old[k]) map_put_raw(m, old[k]);
    free(old);
  }
  b = (Block*)calloc(1, sizeof(Block));
  b->idx = i; b->stamp_view = -1; b->stamp_esdf = -1; b->stamp_cam = -1;
  map_put_raw(m, b);
  if (m->count + 1 > m->order_cap) {
    m->order_cap = m->order_cap ? m->order_cap * 2 : 1024;
    m->order = (Block**)realloc(m->order, (size_t)m->order_cap * sizeof(Block*));
  }
  m->order[m->count++] = b;
  return b;
}
static void block_free(Block* b) {
  free(b->tsdf); free(b->color); free(b->esdf); free(b->fs);
  free(b->mesh.vert); free(b->mesh.nrm); free(b->mesh.col); free(b->mesh.tri);
  free(b);
}
static void map_rebuild(OrcMap* m) { /* after removals: rebuild table from order[] */
  memset(m->table, 0, (size_t)m->cap * sizeof(Block*));
  for (int64_t k = 0; k < m->count; k++) map_put_raw(m, m->order[k]);
}
static void ensure_layer(Block* b, uint32_t layer) {
  if (b->flags & layer) return;
  if (layer == L_TSDF) b->tsdf = (TsdfVoxel*)calloc(NVOX, sizeof(TsdfVoxel));
  if (layer == L_COLOR) b->color = (ColorVoxel*)calloc(NVOX, sizeof(ColorVoxel));
  if (layer == L_ESDF) b->esdf = (EsdfVoxel*)calloc(NVOX, sizeof(EsdfVoxel));
  if (layer == L_FREESPACE) b->fs = (FreespaceVoxel*)calloc(NVOX, sizeof(FreespaceVoxel));
  b->flags |= layer;
}

OrcMap* orc_create(const OrcParams* p) {
  OrcMap* m = (OrcMap*)calloc(1, sizeof(OrcMap));
  m->p = *p;
  return m;
}
void orc_set_params(OrcMap* m, const OrcParams* p) { m->p = *p; }
void orc_destroy(OrcMap* m) {
  if (!m) return;
  for (int64_t k = 0; k < m->count; k++) block_free(m->order[k]);
  free(m->table); free(m->order); free(m->view); free(m->cview); free(m->synth); free(m->cleared); free(m);
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ geometry */
typedef struct { float r[9]; float t[3]; } Rt;   /* p' = R p + t, R row-major */

/* T is a row-major 4x4 rigid transform T_L_C.  Inverse = (R^T, -(R^T t)). */
static void rt_from_T(const float* T, Rt* fwd, Rt* inv) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) fwd->r[3 * i + j] = T[4 * i + j]; fwd->t[i] = T[4 * i + 3]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv->r[3 * i + j] = fwd->r[3 * j + i];
  for (int i = 0; i < 3; i++) {
    float s = inv->r[3 * i + 0] * fwd->t[0];
    s = s + inv->r[3 * i + 1] * fwd->t[1];
    s = s + inv->r[3 * i + 2] * fwd->t[2];
    inv->t[i] = -s;
  }
}
static inline void rt_apply(const Rt* a, float x, float y, float z, float* o) {
  for (int i = 0; i < 3; i++) {
    float s = a->r[3 * i + 0] * x;
    s = s + a->r[3 * i + 1] * y;
    s = s + a->r[3 * i + 2] * z;
    o[i] = s + a->t[i];
  }
}
static inline void rt_rotate(const Rt* a, float x, float y, float z, float* o) {
  for (int i = 0; i < 3; i++) {
    float s = a->r[3 * i + 0] * x;
    s = s + a->r[3 * i + 1] * y;
    s = s + a->r[3 * i + 2] * z;
    o[i] = s;
  }
}
static inline int32_t floor_div8(int32_t v) { return v >> 3; }          /* arithmetic shift = floor division */
static inline int32_t mod8(int32_t v) { return v & 7; }
static inline float voxel_center(int32_t bi, int32_t vi, float bs, float vs) {
  /* layer_publishing.cpp:527: block_index * block_size + x * voxel_size + voxel_size / 2.f */
  return ((float)bi * bs + (float)vi * vs) + vs * 0.5f;
}

/* The TSDF / occupancy integrators' voxel centre in the sensor frame: the block's origin is transformed once, the offset of the
 * voxel's centre inside the block is rotated, and the two are added (the kernels evaluate exactly these two terms: the first is
 * uniform per block, the second per voxel position -- csrc/nvbx_internal.h sensor_block_origin / sensor_voxel_offset). */
static inline void voxel_in_sensor(const Rt* T_C_L, int32_t bx, int32_t by, int32_t bz, int x, int y, int z, float bs, float vs, float* pc) {
  float org[3], off[3];
  const float h = vs * 0.5f;
  rt_apply(T_C_L, (float)bx * bs, (float)by * bs, (float)bz * bs, org);
  rt_rotate(T_C_L, (float)x * vs + h, (float)y * vs + h, (float)z * vs + h, off);
  pc[0] = org[0] + off[0]; pc[1] = org[1] + off[1]; pc[2] = org[2] + off[2];
}

typedef struct { float fu, fv, cu, cv; int32_t w, h; } Cam;
static Cam cam_from(const float* c) { Cam k = {c[0], c[1], c[2], c[3], (int32_t)c[4], (int32_t)c[5]}; return k; }

/* Camera::project ([U] nvblox sensors/camera: z<=0 fails, image-plane coords corner-referenced,
 * in view iff 0<=u<=w, 0<=v<=h) */
static inline int cam_project(const Cam* k, const float* p, float* u, float* v) {
  if (p[2] <= 0.0f) return 0;
  *u = k->fu * (p[0] / p[2]) + k->cu;
  *v = k->fv * (p[1] / p[2]) + k->cv;
  if (*u < 0.0f || *v < 0.0f || *u > (float)k->w || *v > (float)k->h) return 0;
  return 1;
}

/* interpolate2DLinear with FloatPixelGreaterThanZero ([U] nvblox interpolation_2d): u,v corner-referenced;
 * subtract 0.5 to get centre-referenced; low pixel = floor; needs low+1 in bounds; all four > 0. */
/* returns 1 = value, 0 = no sample (outside the image), -1 = a depth tap is invalid (<= 0) */
static inline int interp_depth(const float* img, int rows, int cols, float u, float v, int nearest, float* out) {
  if (nearest) {
    int c = (int)floorf(u), r = (int)floorf(v);
    if (c < 0 || r < 0 || c >= cols || r >= rows) return 0;
    float d = img[(int64_t)r * cols + c];
    if (!(d > 0.0f)) return -1;
    *out = d; return 1;
  }
  float uc = u - 0.5f, vc = v - 0.5f;
  float fx = floorf(uc), fy = floorf(vc);
  int x0 = (int)fx, y0 = (int)fy;
  if (x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1) return 0;
  float ax = uc - fx, ay = vc - fy;
  float f00 = img[(int64_t)y0 * cols + x0], f10 = img[(int64_t)y0 * cols + x0 + 1];
  float f01 = img[(int64_t)(y0 + 1) * cols + x0], f11 = img[(int64_t)(y0 + 1) * cols + x0 + 1];
  if (!(f00 > 0.0f) || !(f10 > 0.0f) || !(f01 > 0.0f) || !(f11 > 0.0f)) return -1;
  float top = (1.0f - ax) * f00 + ax * f10;
  float bot = (1.0f - ax) * f01 + ax * f11;
  *out = (1.0f - ay) * top + ay * bot;
  return 1;
}

/* WeightingFunction (mapper_initialization.cpp:31-42 names the six modes).  constant and inverse-square are unambiguous; the
 * other four formulas are [U]/[D] and exist in two sets (OrcParams.tsdf_weighting_variant):
 *   A: dropoff = linear ramp 1 -> 0 from the surface to -trunc behind it; tsdf-distance penalty = trunc / sdf for voxels more than
 *      trunc in front of the surface; linear-with-max = min(1, 1 / d)
 *   B: dropoff starts one voxel behind the surface, (trunc + sdf) / (trunc - voxel); tsdf-distance penalty = ((trunc + sdf) / trunc)^2
 *      behind the surface; linear-with-max = max(0.01, 1 - d / max_integration_distance) */
static inline float weight_fn(int mode, int variant, float d_meas, float d_vox, float trunc, float voxel_size, float max_dist) {
  float w = 1.0f;
  if (mode == W_INVERSE_SQUARE || mode == W_INVERSE_SQUARE_DROPOFF || mode == W_INVERSE_SQUARE_TSDF_DISTANCE_PENALTY) {
    w = 1.0f / (d_meas * d_meas);
  } else if (mode == W_LINEAR_WITH_MAX) {
    if (variant == 0) { w = 1.0f / d_meas; if (w > 1.0f) w = 1.0f; }
    else { w = 1.0f - d_meas / max_dist; if (w < 0.01f) w = 0.01f; }
  }
  const float sdf = d_meas - d_vox;
  if (mode == W_CONSTANT_DROPOFF || mode == W_INVERSE_SQUARE_DROPOFF) {
    if (variant == 0) {
      if (sdf < 0.0f) { float f = (trunc + sdf) / trunc; if (f < 0.0f) f = 0.0f; w = w * f; }
    } else if (sdf < -voxel_size) {
      float f = 0.0f;
      if (trunc > voxel_size) { f = (trunc + sdf) / (trunc - voxel_size); if (f < 0.0f) f = 0.0f; }
      w = w * f;
    }
  } else if (mode == W_INVERSE_SQUARE_TSDF_DISTANCE_PENALTY) {
    if (variant == 0) { if (sdf > trunc) w = w * (trunc / sdf); }
    else if (sdf < 0.0f) { float f = (trunc + sdf) / trunc; if (f < 0.0f) f = 0.0f; w = w * (f * f); }
  }
  return w;
}
/* [U] UpdateTsdfVoxelFunctor restated, with the open choices as switches: the voxel exactly at sdf == -trunc
 * (tsdf_skip_at_negative_truncation) and the max_weight clamp order (tsdf_weight_clamp_before_blend).  `max_dist` = the sensor's
 * max integration distance (camera / LiDAR).  Returns 1 if the voxel was updated. */
static inline int tsdf_fuse(const OrcParams* p, TsdfVoxel* vx, float ds, float vd, float trunc, float max_dist) {
  const float sdf = ds - vd;
  if (p->tsdf_skip_at_negative_truncation ? (sdf <= -trunc) : (sdf < -trunc)) return 0;
  const float wm = weight_fn(p->weighting_mode, p->tsdf_weighting_variant, ds, vd, trunc, p->voxel_size, max_dist);
  const float wsum = wm + vx->weight;
  if (!(wsum > 0.0f)) return 0;
  float fused; const float wnew = fminf(wsum, p->max_weight);
  if (!p->tsdf_weight_clamp_before_blend) fused = (sdf * wm + vx->distance * vx->weight) / wsum;
  else { float wp = wnew - wm; if (wp < 0.0f) wp = 0.0f; fused = (sdf * wm + vx->distance * wp) / (wm + wp); }
  if (fused > 0.0f) fused = fminf(trunc, fused); else fused = fmaxf(-trunc, fused);
  vx->distance = fused; vx->weight = wnew;
  return 1;
}

/* ------------------------------------------------------------------ view calculation */
/* [U] ViewCalculator::getBlocksInImageViewRaycast restated: one ray per subsampled pixel (overhang clamped to the
 * border), end point at depth+trunc (clipped to max integration distance), Amanatides-Woo walk through the block grid. */
/* [U] workspace bounds of the view calculator: a block is kept iff its cube overlaps the bounds (height bounds: the z
 * interval only; bounding box: all three axes); unbounded keeps everything. */
static int block_in_workspace(const OrcParams* p, Idx3 i) {
  if (p->workspace_bounds_type == 0) return 1;
  const float bs = p->voxel_size * 8.0f;
  const int32_t idx[3] = {i.x, i.y, i.z};
  for (int a = (p->workspace_bounds_type == 1 ? 2 : 0); a < 3; a++) {
    const float lo = (float)idx[a] * bs, hi = (float)(idx[a] + 1) * bs;
    if (!(hi > p->workspace_bounds_min_corner_m[a]) || !(lo < p->workspace_bounds_max_corner_m[a])) return 0;
  }
  return 1;
}
static void view_push(OrcMap* m, Idx3 i) {
  if (!block_in_workspace(&m->p, i)) return;
  Block* b = map_get_or_create(m, i);
  if (b->stamp_view == m->frame) return;
  b->stamp_view = m->frame;
  if (m->camera_frame == m->frame) b->stamp_cam = m->frame;     /* (camera entry points set camera_frame = frame before the view calculation) */
  ensure_layer(b, L_TSDF);
  b->dirty_esdf = 1; b->dirty_mesh = 1;
  if (m->n_view + 1 > m->view_cap) { m->view_cap = m->view_cap ? m->view_cap * 2 : 1024; m->view = (Idx3*)realloc(m->view, (size_t)m->view_cap * sizeof(Idx3)); }
  m->view[m->n_view++] = i;
}

/* [U] Amanatides-Woo through the block grid.  The crossing parameters are evaluated in CLOSED FORM -- crossing number k of axis a at
 * T_a(k) = fmaf(k, tdelta_a, tmax0_a), one rounding -- instead of accumulated by k additions: the traversal then depends on the crossing counts
 * alone, which lets the kernel enter a ray at any step without replaying the steps before it (csrc/tsdf.hip dda_step / dda_jump, same fmaf). */
/* 0 (default) = crossing parameters in closed form, T_a(k) = fmaf(k, tdelta_a, tmax0_a) -- the definition shared with the product;
 * 1 = the textbook Amanatides-Woo accumulation tmax_a += tdelta_a.  Test infrastructure only (tests/test_independent_checks.py counts the blocks
 * on which the two disagree: near-ties at ulp level, nothing else). */
static int g_traversal_accumulate = 0;
void orc_set_traversal_accumulate(int on) { g_traversal_accumulate = on ? 1 : 0; }
static void raycast_blocks(OrcMap* m, const float* o, const float* e, float bs) {
  float s[3], t[3];
  int32_t cur[3], end[3], step[3], ncross[3] = {0, 0, 0};
  float tmax[3], tdelta[3], tmax0[3];
  int32_t nsteps = 0;
  for (int a = 0; a < 3; a++) {
    s[a] = o[a] / bs; t[a] = e[a] / bs;
    cur[a] = (int32_t)floorf(s[a]); end[a] = (int32_t)floorf(t[a]);
    int32_t d = end[a] - cur[a]; nsteps += d < 0 ? -d : d;
    float ray = t[a] - s[a];
    step[a] = ray > 0.0f ? 1 : (ray < 0.0f ? -1 : 0);
    float corrected = step[a] > 0 ? 1.0f : 0.0f;
    float dist_to_boundary = corrected - (s[a] - (float)cur[a]);
    if (fabsf(ray) < 1e-9f) { tmax[a] = 2.0f; tdelta[a] = 2.0f; }
    else { tmax[a] = dist_to_boundary / ray; tdelta[a] = (float)step[a] / ray; }
    tmax0[a] = tmax[a];
  }
  for (int32_t k = 0; k <= nsteps; k++) {
    Idx3 i = {cur[0], cur[1], cur[2]};
    view_push(m, i);
    int a = 0;
    if (tmax[1] < tmax[a]) a = 1;
    if (tmax[2] < tmax[a]) a = 2;
    cur[a] += step[a];
    ncross[a]++;
    tmax[a] = g_traversal_accumulate ? tmax[a] + tdelta[a] : fmaf((float)ncross[a], tdelta[a], tmax0[a]);
  }
}

static void view_calc(OrcMap* m, const float* depth, int rows, int cols, const Rt* T_L_C, const Cam* k) {
  const OrcParams* p = &m->p;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float trunc = p->truncation_distance_vox * vs;
  const int f = p->raycast_subsampling_factor < 1 ? 1 : p->raycast_subsampling_factor;
  m->n_view = 0;
  const float* o = T_L_C->t;
  /* ray grid: indices i with i*f < rows + f - 1 ; pixel = min(i*f, rows-1) */
  for (int ri = 0; ri * f < rows + f - 1; ri++) {
    int prow = ri * f; if (prow >= rows) prow = rows - 1;
    for (int ci = 0; ci * f < cols + f - 1; ci++) {
      int pcol = ci * f; if (pcol >= cols) pcol = cols - 1;
      float d = depth[(int64_t)prow * cols + pcol];
      if (!(d > 0.0f)) continue;
      float de = d + trunc;
      if (p->max_integration_distance_m > 0.0f && de > p->max_integration_distance_m) de = p->max_integration_distance_m;
      /* vectorFromPixelIndices: pixel centre at +0.5 */
      float rx = (((float)pcol + 0.5f) - k->cu) / k->fu;
      float ry = (((float)prow + 0.5f) - k->cv) / k->fv;
      float pc[3] = {de * rx, de * ry, de};
      float pl[3];
      rt_apply(T_L_C, pc[0], pc[1], pc[2], pl);
      raycast_blocks(m, o, pl, bs);
    }
  }
}

static float log_odds(float p) { return logf(p / (1.0f - p)); }

/* ------------------------------------------------------------------ TSDF */
/* [U] ProjectiveTsdfIntegrator::integrateFrame -> integrateBlocksKernel + UpdateTsdfVoxelFunctor restated. */
static void tsdf_integrate_block(const OrcParams* p, Block* b, const float* depth, int rows, int cols, const Rt* T_C_L, const Cam* k) {
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float trunc = p->truncation_distance_vox * vs;
  for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
    float pc[3]; voxel_in_sensor(T_C_L, b->idx.x, b->idx.y, b->idx.z, x, y, z, bs, vs, pc);
    float u, v;
    if (!cam_project(k, pc, &u, &v)) continue;
    const float vd = pc[2];
    if (p->max_integration_distance_m > 0.0f && vd > p->max_integration_distance_m) continue;
    float ds;
    TsdfVoxel* vx = &b->tsdf[z + 8 * y + 64 * x];
    const int got = interp_depth(depth, rows, cols, u, v, p->depth_interp_nearest, &ds);
    if (p->projective_layer_type == 1) {
      /* [U] ProjectiveOccupancyIntegrator restated: log-odds update by region along the ray (free in front of the measured
       * surface, occupied within +-occupied_region_half_width_m of it, "unobserved" behind), clamped to +-10.  The log-odds
       * live in TsdfVoxel.distance of the projective layer; weight stays 0. */
      if (got > 0) {
        float upd = log_odds(p->unobserved_region_occupancy_probability);
        if (vd < ds - p->occupied_region_half_width_m) upd = log_odds(p->free_region_occupancy_probability);
        else if (vd <= ds + p->occupied_region_half_width_m) upd = log_odds(p->occupied_region_occupancy_probability);
        float v = vx->distance + upd;
        if (v > 10.0f) v = 10.0f;
        if (v < -10.0f) v = -10.0f;
        vx->distance = v; vx->weight = 0.0f;
      }
      continue;
    }
    if (got < 0 && p->invalid_depth_decay_factor >= 0.0f) {
      /* [U] invalid depth where the voxel projects: the surface estimate there loses confidence */
      vx->weight = vx->weight * p->invalid_depth_decay_factor;
      continue;
    }
    if (got <= 0) continue;
    tsdf_fuse(p, vx, ds, vd, trunc, p->max_integration_distance_m);
  }
}

/* [U] DepthPreprocessor (do_depth_preprocessing / depth_preprocessing_num_dilations, mapper_initialization.cpp:238-243):
 * regions of invalid depth (<= 0) are dilated by n pixels (n 3x3 dilations = a (2n+1)^2 square) before integration, which
 * removes the unreliable rim of pixels around depth holes. */
static float* dilate_invalid(const float* depth, int rows, int cols, int n) {
  float* out = (float*)malloc(sizeof(float) * (size_t)rows * cols);
#pragma omp parallel for
  for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) {
    int bad = 0;
    for (int dr = -n; dr <= n && !bad; dr++) for (int dc = -n; dc <= n; dc++) {
      const int rr = r + dr, cc = c + dc;
      if (rr < 0 || cc < 0 || rr >= rows || cc >= cols) continue;
      if (!(depth[(int64_t)rr * cols + cc] > 0.0f)) { bad = 1; break; }
    }
    out[(int64_t)r * cols + c] = bad ? 0.0f : depth[(int64_t)r * cols + c];
  }
  return out;
}

/* ------------------------------------------------------------------ freespace layer (dynamic mapping) */
/* [U] FreespaceIntegrator::updateFreespaceLayer restated (MappingType::kDynamic; parameters mapper_initialization.cpp:430-462,
 * values nvblox_dynamics.yaml:12-18).  Every voxel of the blocks in the depth view, after the TSDF update, at time `now`:
 *   first touch: last_occupied = now, duration 0, high confidence = initialize_to_high_confidence_freespace;
 *   observed (w > 0) and occupied (d < max_tsdf_distance_for_occupancy_m; with check_neighborhood also if a 6-neighbour is):
 *     the consecutive occupancy duration grows by the gap since the last occupied observation if that gap is at most
 *     max_unobserved_to_keep_consecutive_occupancy_ms, else restarts at 0; after min_consecutive_occupancy_duration_for_reset_ms
 *     of it the voxel is no longer high-confidence freespace (it has become part of the static world);
 *   observed and free: high-confidence freespace once min_duration_since_occupied_for_freespace_ms have passed since it was
 *     last occupied (a moving object passing through does not reset it -- that is what makes it detectable). */
static int tsdf_occupied(const OrcMap* m, int32_t gx, int32_t gy, int32_t gz) {
  Idx3 bi = {floor_div8(gx), floor_div8(gy), floor_div8(gz)};
  const Block* b = map_find(m, bi);
  if (!b || !(b->flags & L_TSDF)) return 0;
  const TsdfVoxel* v = &b->tsdf[mod8(gz) + 8 * mod8(gy) + 64 * mod8(gx)];
  return v->weight > 0.0f && v->distance < m->p.max_tsdf_distance_for_occupancy_m;
}
static void freespace_update(OrcMap* m) {
  const OrcParams* p = &m->p;
  const int64_t now = m->time_ms;
  for (int64_t i = 0; i < m->n_view; i++) {
    Block* b = map_find(m, m->view[i]);
    if (!b || !(b->flags & L_TSDF)) continue;
    ensure_layer(b, L_FREESPACE);
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      const int li = z + 8 * y + 64 * x;
      FreespaceVoxel* f = &b->fs[li];
      if (!f->initialized) { f->initialized = 1; f->last_occupied_timestamp_ms = now; f->consecutive_occupancy_duration_ms = 0;
                             f->is_high_confidence_freespace = p->initialize_to_high_confidence_freespace ? 1 : 0; }
      const TsdfVoxel* tv = &b->tsdf[li];
      if (!(tv->weight > 0.0f)) continue;
      const int32_t gx = b->idx.x * 8 + x, gy = b->idx.y * 8 + y, gz = b->idx.z * 8 + z;
      int occupied = tv->distance < p->max_tsdf_distance_for_occupancy_m;
      if (!occupied && p->check_neighborhood)
        occupied = tsdf_occupied(m, gx - 1, gy, gz) || tsdf_occupied(m, gx + 1, gy, gz) || tsdf_occupied(m, gx, gy - 1, gz) ||
                   tsdf_occupied(m, gx, gy + 1, gz) || tsdf_occupied(m, gx, gy, gz - 1) || tsdf_occupied(m, gx, gy, gz + 1);
      if (occupied) {
        const int64_t gap = now - f->last_occupied_timestamp_ms;
        f->consecutive_occupancy_duration_ms = gap <= (int64_t)p->max_unobserved_to_keep_consecutive_occupancy_ms ? f->consecutive_occupancy_duration_ms + gap : 0;
        f->last_occupied_timestamp_ms = now;
        if (f->consecutive_occupancy_duration_ms >= (int64_t)p->min_consecutive_occupancy_duration_for_reset_ms) f->is_high_confidence_freespace = 0;
      } else if (now - f->last_occupied_timestamp_ms >= (int64_t)p->min_duration_since_occupied_for_freespace_ms) {
        f->is_high_confidence_freespace = 1;
      }
    }
  }
}
void orc_set_time_ms(OrcMap* m, int64_t t) { m->time_ms = t; }

/* [U] DynamicsDetection::computeDynamics restated: a valid depth pixel (within max_distance) whose 3-D point lies in a
 * high-confidence-freespace voxel is dynamic. */
void orc_detect_dynamics(const OrcMap* m, const float* depth, int32_t rows, int32_t cols, const float* T_L_C16, const float* cam6, float max_distance_m, uint8_t* mask) {
  Rt T_L_C, T_C_L; rt_from_T(T_L_C16, &T_L_C, &T_C_L);
  const Cam k = cam_from(cam6);
  const float vs = m->p.voxel_size;
  for (int32_t r = 0; r < rows; r++) for (int32_t c = 0; c < cols; c++) {
    const int64_t i = (int64_t)r * cols + c;
    mask[i] = 0;
    const float d = depth[i];
    if (!(d > 0.0f) || (max_distance_m > 0.0f && d > max_distance_m)) continue;
    const float rx = (((float)c + 0.5f) - k.cu) / k.fu, ry = (((float)r + 0.5f) - k.cv) / k.fv;
    float pl[3]; rt_apply(&T_L_C, d * rx, d * ry, d, pl);
    const int32_t gx = (int32_t)floorf(pl[0] / vs), gy = (int32_t)floorf(pl[1] / vs), gz = (int32_t)floorf(pl[2] / vs);
    Idx3 bi = {floor_div8(gx), floor_div8(gy), floor_div8(gz)};
    const Block* b = map_find(m, bi);
    if (!b || !(b->flags & L_FREESPACE)) continue;
    if (b->fs[mod8(gz) + 8 * mod8(gy) + 64 * mod8(gx)].is_high_confidence_freespace) mask[i] = 1;
  }
}
/* [U] removeSmallConnectedComponents (multi_mapper.connected_mask_component_size_threshold, mapper_initialization.cpp:130): 8-connected
 * components of the non-zero mask pixels smaller than min_size are erased. */
void orc_remove_small_components(uint8_t* mask, int32_t rows, int32_t cols, int32_t min_size) {
  const int64_t n = (int64_t)rows * cols;
  int32_t* label = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  int64_t* stack = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) label[i] = -1;
  for (int64_t s0 = 0; s0 < n; s0++) {
    if (!mask[s0] || label[s0] >= 0) continue;
    int64_t top = 0, size = 0; stack[top++] = s0; label[s0] = (int32_t)s0;
    int64_t first = 0;
    (void)first;
    /* first pass: flood fill, count */
    int64_t head = 0;
    while (head < top) {
      const int64_t i = stack[head++]; size++;
      const int32_t r = (int32_t)(i / cols), c = (int32_t)(i % cols);
      for (int dr = -1; dr <= 1; dr++) for (int dc = -1; dc <= 1; dc++) {
        if (!dr && !dc) continue;
        const int32_t rr = r + dr, cc = c + dc;
        if (rr < 0 || cc < 0 || rr >= rows || cc >= cols) continue;
        const int64_t j = (int64_t)rr * cols + cc;
        if (mask[j] && label[j] < 0) { label[j] = (int32_t)s0; stack[top++] = j; }
      }
    }
    if (size < min_size) for (int64_t q = 0; q < top; q++) mask[stack[q]] = 0;
  }
  free(label); free(stack);
}

int64_t orc_integrate_depth(OrcMap* m, const float* depth_in, int rows, int cols, const float* T_L_C16, const float* cam6) {
  float* pre = NULL;
  if (m->p.do_depth_preprocessing && m->p.depth_preprocessing_num_dilations > 0) pre = dilate_invalid(depth_in, rows, cols, m->p.depth_preprocessing_num_dilations);
  const float* depth = pre ? pre : depth_in;
  Rt T_L_C, T_C_L; rt_from_T(T_L_C16, &T_L_C, &T_C_L);
  Cam k = cam_from(cam6);
  m->frame++; m->camera_frame = m->frame;
  view_calc(m, depth, rows, cols, &T_L_C, &k);
  const int64_t n = m->n_view;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < n; i++) {
    Block* b = map_find(m, m->view[i]);
    tsdf_integrate_block(&m->p, b, depth, rows, cols, &T_C_L, &k);
  }
  if (m->p.projective_layer_type == 2) freespace_update(m);
  free(pre);
  return n;
}


/* ------------------------------------------------------------------ multi-GPU measurement exchange (tests/test_dist_gloo.py)
 * The CPU counterpart of nvbx_measure_depth / nvbx_apply_measurements (include/nvblox_hip.h): records {x, y, z, rank, 512 x {ds, vd}}. */
typedef struct { int32_t x, y, z, rank; float v[512][2]; } MeasRec;
int64_t orc_measure_depth(OrcMap* m, const float* depth, int rows, int cols, const float* T_L_C16, const float* cam6, MeasRec* out, int64_t cap) {
  Rt T_L_C, T_C_L; rt_from_T(T_L_C16, &T_L_C, &T_C_L);
  Cam k = cam_from(cam6);
  const OrcParams* p = &m->p;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  m->frame++; m->camera_frame = m->frame;
  view_calc(m, depth, rows, cols, &T_L_C, &k);
  /* (view_calc marks the blocks dirty; their values arrive with orc_apply_measurements) */
  int64_t n = m->n_view < cap ? m->n_view : cap;
  for (int64_t i = 0; i < n; i++) {
    const Idx3 bi = m->view[i];
    MeasRec* r = &out[i];
    r->x = bi.x; r->y = bi.y; r->z = bi.z; r->rank = 0;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      float pc[3]; voxel_in_sensor(&T_C_L, bi.x, bi.y, bi.z, x, y, z, bs, vs, pc);
      float u, v, ds = 0.0f;
      float* o = r->v[z + 8 * y + 64 * x];
      o[0] = 0.0f; o[1] = -1.0f;
      if (!cam_project(&k, pc, &u, &v)) continue;
      const float vd = pc[2];
      if (p->max_integration_distance_m > 0.0f && vd > p->max_integration_distance_m) continue;
      const int got = interp_depth(depth, rows, cols, u, v, p->depth_interp_nearest, &ds);
      if (got > 0) { o[0] = ds; o[1] = vd; } else if (got < 0) { o[0] = -1.0f; o[1] = vd; }
    }
  }
  return n;
}
int64_t orc_apply_measurements(OrcMap* m, const MeasRec* all, const int32_t* counts, int32_t world, int64_t stride, int32_t owner_mod, int32_t owner_rank) {
  const OrcParams* p = &m->p;
  const float trunc = p->truncation_distance_vox * p->voxel_size;
  m->frame++; m->camera_frame = m->frame;
  m->n_view = 0;
  int64_t applied = 0;
  for (int32_t r = 0; r < world; r++) {
    int64_t n = counts[r]; if (n > stride) n = stride;
    for (int64_t i = 0; i < n; i++) {
      const MeasRec* rec = &all[(int64_t)r * stride + i];
      Idx3 bi = {rec->x, rec->y, rec->z};
      if (owner_mod > 1 && (int32_t)(idx_hash(bi) % (uint32_t)owner_mod) != owner_rank) continue;
      view_push(m, bi);                                   /* allocates, stamps the view, dirties the block */
      Block* b = map_find(m, bi);
      if (!b) continue;
      ensure_layer(b, L_TSDF); b->dirty_esdf = 1; b->dirty_mesh = 1;
      for (int q = 0; q < NVOX; q++) {
        const float ds = rec->v[q][0], vd = rec->v[q][1];
        if (vd < 0.0f) continue;
        TsdfVoxel* vx = &b->tsdf[q];
        if (p->projective_layer_type == 1) {
          if (ds < 0.0f) continue;
          float upd = log_odds(p->unobserved_region_occupancy_probability);
          if (vd < ds - p->occupied_region_half_width_m) upd = log_odds(p->free_region_occupancy_probability);
          else if (vd <= ds + p->occupied_region_half_width_m) upd = log_odds(p->occupied_region_occupancy_probability);
          float v = vx->distance + upd; if (v > 10.0f) v = 10.0f; if (v < -10.0f) v = -10.0f;
          vx->distance = v; vx->weight = 0.0f;
        } else if (ds < 0.0f) { if (p->invalid_depth_decay_factor >= 0.0f) vx->weight = vx->weight * p->invalid_depth_decay_factor; }
        else tsdf_fuse(p, vx, ds, vd, trunc, p->max_integration_distance_m);
      }
      applied++;
    }
  }
  return applied;
}

/* ------------------------------------------------------------------ LiDAR (range image) */
/* [U] ProjectiveTsdfIntegrator::integrateFrame(DepthImage, T_L_C, Lidar) restated (call site nvblox_node.cpp:1382-1384;
 * model anchors in nvbx_lidar_math.h).  Same view calculation and voxel update as the camera path with the sensor
 * model swapped: ray through the pixel centre = beam direction; voxel depth = range; measured depth by
 * interpolateLidarImage ([U]: bilinear if the 4 beams are valid and agree within max_allowable_difference, else the
 * nearest beam if the voxel centre is within max_allowable_dist_to_ray of it). */
typedef struct { nvbx_lidar_model l; float* el; float* az; } LidarTab;   /* {sin, cos} pairs */
static LidarTab lidar_tab_make(const float* lidar5) {
  LidarTab t;
  t.l = nvbx_lidar_make((int32_t)lidar5[0], (int32_t)lidar5[1], lidar5[2], lidar5[3], lidar5[4]);
  t.el = (float*)malloc(sizeof(float) * 2 * (size_t)t.l.rows); t.az = (float*)malloc(sizeof(float) * 2 * (size_t)t.l.cols);
  for (int k = 0; k < t.l.rows; k++) { const double el = (double)t.l.max_el - (double)k * (double)t.l.rpp_el; t.el[2 * k] = (float)sin(el); t.el[2 * k + 1] = (float)cos(el); }
  for (int j = 0; j < t.l.cols; j++) { const double az = -(double)NVBX_PI_F + (double)j * (double)t.l.rpp_az; t.az[2 * j] = (float)sin(az); t.az[2 * j + 1] = (float)cos(az); }
  return t;
}
static inline void lidar_dir(const LidarTab* t, int row, int col, float* d) {
  const float se = t->el[2 * row], ce = t->el[2 * row + 1], sa = t->az[2 * col], ca = t->az[2 * col + 1];
  d[0] = ce * ca; d[1] = ce * sa; d[2] = se;
}
static int lidar_sample(const OrcParams* p, const LidarTab* t, const float* img, int rows, int cols, const float* pc, float max_dist, float* ds, float* vd) {
  const float r = nvbx_lidar_range(pc);
  float u, v;
  if (!nvbx_lidar_project(&t->l, pc, r, &u, &v)) return 0;
  *vd = r;
  if (max_dist > 0.0f && r > max_dist) return 0;
  const float max_diff = p->lidar_linear_interpolation_max_allowable_difference_vox * p->voxel_size;
  const float max_ray = p->lidar_nearest_interpolation_max_allowable_dist_to_ray_vox * p->voxel_size;
  const float uc = u - 0.5f, vc = v - 0.5f;
  const float fx = floorf(uc), fy = floorf(vc);
  const int x0 = (int)fx, y0 = (int)fy;
  if (!(x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1)) {
    const float f00 = img[(int64_t)y0 * cols + x0], f10 = img[(int64_t)y0 * cols + x0 + 1];
    const float f01 = img[(int64_t)(y0 + 1) * cols + x0], f11 = img[(int64_t)(y0 + 1) * cols + x0 + 1];
    if (f00 > 0.0f && f10 > 0.0f && f01 > 0.0f && f11 > 0.0f) {
      const float mx = fmaxf(fmaxf(f00, f10), fmaxf(f01, f11)), mn = fminf(fminf(f00, f10), fminf(f01, f11));
      if (mx - mn <= max_diff) {
        const float ax = uc - fx, ay = vc - fy;
        const float top = fmaf(ax, f10, (1.0f - ax) * f00);          /* fused multiply-adds written out (one rounding each), as in the kernel */
        const float bot = fmaf(ax, f11, (1.0f - ax) * f01);
        *ds = fmaf(ay, bot, (1.0f - ay) * top);
        return 1;
      }
    }
  }
  const int c = (int)floorf(u), rr = (int)floorf(v);
  if (c < 0 || rr < 0 || c >= cols || rr >= rows) return 0;
  const float d = img[(int64_t)rr * cols + c];
  if (!(d > 0.0f)) return 0;
  float dir[3]; lidar_dir(t, rr, c, dir);
  const float dot = fmaf(pc[2], dir[2], fmaf(pc[1], dir[1], pc[0] * dir[0]));
  const float ex = fmaf(-dot, dir[0], pc[0]), ey = fmaf(-dot, dir[1], pc[1]), ez = fmaf(-dot, dir[2], pc[2]);
  if (fmaf(ez, ez, fmaf(ey, ey, ex * ex)) > max_ray * max_ray) return 0;       /* squared point-to-ray distance against the squared threshold */
  *ds = d;
  return 1;
}

int64_t orc_integrate_lidar_depth(OrcMap* m, const float* range, int rows, int cols, const float* T_L_C16, const float* lidar5) {
  Rt T_L_C, T_C_L; rt_from_T(T_L_C16, &T_L_C, &T_C_L);
  const OrcParams* p = &m->p;
  LidarTab tab = lidar_tab_make(lidar5);
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float trunc = p->truncation_distance_vox * vs;
  const float max_dist = p->lidar_max_integration_distance_m;
  const int f = p->raycast_subsampling_factor < 1 ? 1 : p->raycast_subsampling_factor;
  m->frame++;
  m->n_view = 0;
  for (int ri = 0; ri * f < rows + f - 1; ri++) {
    int prow = ri * f; if (prow >= rows) prow = rows - 1;
    for (int ci = 0; ci * f < cols + f - 1; ci++) {
      int pcol = ci * f; if (pcol >= cols) pcol = cols - 1;
      const float d = range[(int64_t)prow * cols + pcol];
      if (!(d > 0.0f)) continue;
      float de = d + trunc;
      if (max_dist > 0.0f && de > max_dist) de = max_dist;
      float dir[3]; lidar_dir(&tab, prow, pcol, dir);
      float pl[3];
      rt_apply(&T_L_C, de * dir[0], de * dir[1], de * dir[2], pl);
      raycast_blocks(m, T_L_C.t, pl, bs);
    }
  }
  const int64_t n = m->n_view;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < n; i++) {
    Block* b = map_find(m, m->view[i]);
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      float pc[3]; voxel_in_sensor(&T_C_L, b->idx.x, b->idx.y, b->idx.z, x, y, z, bs, vs, pc);
      float ds, vd;
      if (!lidar_sample(p, &tab, range, rows, cols, pc, max_dist, &ds, &vd)) continue;
      tsdf_fuse(p, &b->tsdf[z + 8 * y + 64 * x], ds, vd, trunc, max_dist);
    }
  }
  free(tab.el); free(tab.az);
  return n;
}

/* depthImageFromPointcloudKernel restated (conversions/pointcloud_conversions.cu:118-150); points in order, last writer wins */
void orc_depth_image_from_pointcloud(const float* pts, int64_t n, const float* lidar5, float* img) {
  const nvbx_lidar_model l = nvbx_lidar_make((int32_t)lidar5[0], (int32_t)lidar5[1], lidar5[2], lidar5[3], lidar5[4]);
  memset(img, 0, sizeof(float) * (size_t)l.rows * (size_t)l.cols);
  for (int64_t i = 0; i < n; i++) {
    const float* q = pts + 3 * i;
    if (isnan(q[0]) || isnan(q[1]) || isnan(q[2])) continue;
    const float r = nvbx_lidar_range(q);
    float u, v;
    if (!nvbx_lidar_project(&l, q, r, &u, &v)) continue;
    const int c = (int)floorf(u), rr = (int)floorf(v);
    if (c < 0 || rr < 0 || c >= l.cols || rr >= l.rows) continue;
    img[(int64_t)rr * l.cols + c] = r;
  }
}
/* exported for the model tests: project one point; returns 1 and (u, v) if inside the model */
int orc_lidar_project(const float* lidar5, const float* p, float* u, float* v) {
  const nvbx_lidar_model l = nvbx_lidar_make((int32_t)lidar5[0], (int32_t)lidar5[1], lidar5[2], lidar5[3], lidar5[4]);
  return nvbx_lidar_project(&l, p, nvbx_lidar_range(p), u, v);
}
float orc_atan2f(float y, float x) { return nvbx_atan2f(y, x); }
/* exported for tests/test_independent_checks.py: the measurement model (lidar_sample above = LidarSensor::sample of the product) evaluated
 * at n sensor-frame points: branch[i] = 0 no measurement, 1 four-tap bilinear, 2 nearest beam; ds[i] = the measured range */
void orc_lidar_sample_points(const OrcParams* p, const float* lidar5, const float* range, const float* pts, int64_t n, float max_dist, int32_t* branch, float* ds) {
  LidarTab tab = lidar_tab_make(lidar5);
  const int rows = tab.l.rows, cols = tab.l.cols;
  const float max_diff = p->lidar_linear_interpolation_max_allowable_difference_vox * p->voxel_size;
  for (int64_t i = 0; i < n; i++) {
    float d = 0.0f, vd = 0.0f;
    const int got = lidar_sample(p, &tab, range, rows, cols, pts + 3 * i, max_dist, &d, &vd);
    branch[i] = 0; ds[i] = 0.0f;
    if (!got) continue;
    /* which rule produced it: re-derive the bilinear test's outcome the way lidar_sample does (same float operations) */
    float u = 0.0f, v = 0.0f; nvbx_lidar_project(&tab.l, pts + 3 * i, nvbx_lidar_range(pts + 3 * i), &u, &v);
    const float uc = u - 0.5f, vc = v - 0.5f; const int x0 = (int)floorf(uc), y0 = (int)floorf(vc);
    int bil = 0;
    if (!(x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1)) {
      const float f00 = range[(int64_t)y0 * cols + x0], f10 = range[(int64_t)y0 * cols + x0 + 1], f01 = range[(int64_t)(y0 + 1) * cols + x0], f11 = range[(int64_t)(y0 + 1) * cols + x0 + 1];
      if (f00 > 0.0f && f10 > 0.0f && f01 > 0.0f && f11 > 0.0f) {
        const float mx = fmaxf(fmaxf(f00, f10), fmaxf(f01, f11)), mn = fminf(fminf(f00, f10), fminf(f01, f11));
        bil = (mx - mn <= max_diff);
      }
    }
    branch[i] = bil ? 1 : 2; ds[i] = d;
  }
  free(tab.el); free(tab.az);
}

/* ------------------------------------------------------------------ accessors */
static int idx_cmp(const void* a, const void* b) {
  const Idx3* p = (const Idx3*)a; const Idx3* q = (const Idx3*)b;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  if (p->z != q->z) return p->z < q->z ? -1 : 1;
  return 0;
}
int64_t orc_num_blocks(const OrcMap* m, uint32_t layer) {
  int64_t n = 0;
  for (int64_t k = 0; k < m->count; k++) if (m->order[k]->flags & layer) n++;
  return n;
}
/* sorted (x,y,z) lexicographic */
int64_t orc_block_indices(const OrcMap* m, uint32_t layer, int32_t* out, int64_t cap) {
  int64_t n = 0;
  for (int64_t k = 0; k < m->count; k++) if (m->order[k]->flags & layer) {
    if (n < cap) { out[3 * n] = m->order[k]->idx.x; out[3 * n + 1] = m->order[k]->idx.y; out[3 * n + 2] = m->order[k]->idx.z; }
    n++;
  }
  qsort(out, (size_t)(n < cap ? n : cap), sizeof(Idx3), idx_cmp);
  return n;
}
int64_t orc_last_view(const OrcMap* m, int32_t* out, int64_t cap) {
  int64_t n = m->n_view < cap ? m->n_view : cap;
  memcpy(out, m->view, (size_t)n * sizeof(Idx3));
  qsort(out, (size_t)n, sizeof(Idx3), idx_cmp);
  return m->n_view;
}
int64_t orc_last_color_view(const OrcMap* m, int32_t* out, int64_t cap) {
  int64_t n = m->n_cview < cap ? m->n_cview : cap;
  memcpy(out, m->cview, (size_t)n * sizeof(Idx3));
  qsort(out, (size_t)n, sizeof(Idx3), idx_cmp);
  return m->n_cview;
}
/* copy a block out in the reference's voxel struct layout, index z + 8y + 64x */
int orc_get_block(const OrcMap* m, uint32_t layer, int32_t x, int32_t y, int32_t z, void* out) {
  Idx3 i = {x, y, z};
  Block* b = map_find(m, i);
  if (!b || !(b->flags & layer)) return 0;
  if (layer == L_TSDF) memcpy(out, b->tsdf, NVOX * sizeof(TsdfVoxel));
  else if (layer == L_COLOR) memcpy(out, b->color, NVOX * sizeof(ColorVoxel));
  else if (layer == L_ESDF) memcpy(out, b->esdf, NVOX * sizeof(EsdfVoxel));
  else if (layer == L_FREESPACE) memcpy(out, b->fs, NVOX * sizeof(FreespaceVoxel));
  else return 0;
  return 1;
}
/* allocateBlockAtIndex + voxel write (used by the known-answer test, test_esdf_and_gradient_conversions.cpp:85-92,114) */
int orc_set_block(OrcMap* m, uint32_t layer, int32_t x, int32_t y, int32_t z, const void* in) {
  Idx3 i = {x, y, z};
  Block* b = map_get_or_create(m, i);
  ensure_layer(b, layer);
  if (layer == L_TSDF) { memcpy(b->tsdf, in, NVOX * sizeof(TsdfVoxel)); b->dirty_esdf = 1; b->dirty_mesh = 1; }
  else if (layer == L_COLOR) memcpy(b->color, in, NVOX * sizeof(ColorVoxel));
  else if (layer == L_ESDF) memcpy(b->esdf, in, NVOX * sizeof(EsdfVoxel));
  else return 0;
  return 1;
}

/* ------------------------------------------------------------------ colour */
static inline const TsdfVoxel* tsdf_at_position(const OrcMap* m, const float* pl, float vs) {
  int32_t gx = (int32_t)floorf(pl[0] / vs), gy = (int32_t)floorf(pl[1] / vs), gz = (int32_t)floorf(pl[2] / vs);
  Idx3 bi = {floor_div8(gx), floor_div8(gy), floor_div8(gz)};
  Block* b = map_find(m, bi);
  if (!b || !(b->flags & L_TSDF)) return NULL;
  return &b->tsdf[mod8(gz) + 8 * mod8(gy) + 64 * mod8(gx)];
}

/* [U] SphereTracer::cast restated.  Returns 1 and *t_out on success. */
static int sphere_cast(const OrcMap* m, const float* o, const float* dir, float trunc, float eps_m, int max_steps, float max_len, float* t_out) {
  const float vs = m->p.voxel_size;
  int last_positive = 0;
  float t = 0.0f;
  for (int i = 0; i < max_steps && t < max_len; i++) {
    float pl[3] = {o[0] + t * dir[0], o[1] + t * dir[1], o[2] + t * dir[2]};
    const TsdfVoxel* v = tsdf_at_position(m, pl, vs);
    float step;
    if (!v || !(v->weight > 1e-4f)) {
      if (!last_positive) step = trunc;
      else return 0;
    } else {
      if (v->distance < eps_m) {
        if (last_positive) { *t_out = t + v->distance; return 1; }
        return 0;
      }
      step = v->distance; last_positive = 1;
    }
    t = t + step;
  }
  return 0;
}

static void render_synthetic_depth(OrcMap* m, const Rt* T_L_C, const Cam* k, int rows, int cols) {
  const OrcParams* p = &m->p;
  const int f = p->sphere_tracing_subsampling < 1 ? 1 : p->sphere_tracing_subsampling;
  const int srows = rows / f, scols = cols / f;
  const float trunc = p->truncation_distance_vox * p->voxel_size;
  const float eps_m = p->sphere_tracing_surface_eps_vox * p->voxel_size;
  free(m->synth);
  m->synth = (float*)calloc((size_t)srows * scols, sizeof(float));
  m->synth_rows = srows; m->synth_cols = scols;
#pragma omp parallel for schedule(dynamic, 4)
  for (int r = 0; r < srows; r++) for (int c = 0; c < scols; c++) {
    /* ray through the centre of full-res pixel (c*f, r*f) */
    float rx = (((float)(c * f) + 0.5f) - k->cu) / k->fu;
    float ry = (((float)(r * f) + 0.5f) - k->cv) / k->fv;
    float n = sqrtf((rx * rx + ry * ry) + 1.0f);
    float dc[3] = {rx / n, ry / n, 1.0f / n};
    float dl[3]; rt_rotate(T_L_C, dc[0], dc[1], dc[2], dl);
    float t;
    if (sphere_cast(m, T_L_C->t, dl, trunc, eps_m, p->sphere_tracing_max_steps, p->sphere_tracing_max_ray_length_m, &t))
      m->synth[(int64_t)r * scols + c] = t * dc[2];     /* depth = z component of the hit in C */
  }
}

/* conservative frustum test: reject iff all 8 block corners are outside one of the 6 planes (in camera frame) */
static int block_in_frustum(Idx3 bi, float bs, const Rt* T_C_L, const Cam* k, float max_d) {
  int out[6] = {0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 8; c++) {
    float pl[3] = {(float)(bi.x + (c & 1)) * bs, (float)(bi.y + ((c >> 1) & 1)) * bs, (float)(bi.z + ((c >> 2) & 1)) * bs};
    float pc[3]; rt_apply(T_C_L, pl[0], pl[1], pl[2], pc);
    if (k->fu * pc[0] + k->cu * pc[2] < 0.0f) out[0]++;
    if (k->fu * pc[0] + (k->cu - (float)k->w) * pc[2] > 0.0f) out[1]++;
    if (k->fv * pc[1] + k->cv * pc[2] < 0.0f) out[2]++;
    if (k->fv * pc[1] + (k->cv - (float)k->h) * pc[2] > 0.0f) out[3]++;
    if (pc[2] < 0.0f) out[4]++;
    if (max_d > 0.0f && pc[2] > max_d) out[5]++;
  }
  for (int q = 0; q < 6; q++) if (out[q] == 8) return 0;
  return 1;
}
static int block_in_band(const Block* b, float trunc) {
  for (int i = 0; i < NVOX; i++) if (b->tsdf[i].weight > 1e-4f && fabsf(b->tsdf[i].distance) < trunc) return 1;
  return 0;
}
static inline uint8_t blend_u8(float c0, float w0, float c1, float w1) {
  /* Color::blendTwoColors [U]: normalise weights, blend, round-half-away */
  float tw = w0 + w1;
  float a = w0 / tw, b = w1 / tw;
  float v = c0 * a + c1 * b;
  v = floorf(v + 0.5f);
  if (v < 0.0f) v = 0.0f; if (v > 255.0f) v = 255.0f;
  return (uint8_t)v;
}
static int interp_color(const uint8_t* img, int rows, int cols, float u, float v, float* rgb) {
  float uc = u - 0.5f, vc = v - 0.5f;
  float fx = floorf(uc), fy = floorf(vc);
  int x0 = (int)fx, y0 = (int)fy;
  if (x0 < 0 || y0 < 0 || x0 + 1 > cols - 1 || y0 + 1 > rows - 1) return 0;
  float ax = uc - fx, ay = vc - fy;
  for (int ch = 0; ch < 3; ch++) {
    float f00 = img[((int64_t)y0 * cols + x0) * 3 + ch], f10 = img[((int64_t)y0 * cols + x0 + 1) * 3 + ch];
    float f01 = img[((int64_t)(y0 + 1) * cols + x0) * 3 + ch], f11 = img[((int64_t)(y0 + 1) * cols + x0 + 1) * 3 + ch];
    float top = (1.0f - ax) * f00 + ax * f10;
    float bot = (1.0f - ax) * f01 + ax * f11;
    rgb[ch] = (1.0f - ay) * top + ay * bot;
  }
  return 1;
}

/* [U] ProjectiveColorIntegrator::integrateFrame restated: blocks = allocated TSDF blocks in frustum and in the
 * truncation band; synthetic depth by sphere tracing at 1/f resolution; per voxel occlusion test
 * |synthetic - voxel_depth| <= trunc; bilinear colour; weight 1 blend; weight clamp at max_weight. */
int64_t orc_integrate_color(OrcMap* m, const uint8_t* rgb, int rows, int cols, const float* T_L_C16, const float* cam6) {
  Rt T_L_C, T_C_L; rt_from_T(T_L_C16, &T_L_C, &T_C_L);
  Cam k = cam_from(cam6);
  const OrcParams* p = &m->p;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float trunc = p->truncation_distance_vox * vs;
  const int f = p->sphere_tracing_subsampling < 1 ? 1 : p->sphere_tracing_subsampling;
  render_synthetic_depth(m, &T_L_C, &k, rows, cols);
  m->n_cview = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_TSDF)) continue;
    if (!block_in_frustum(b->idx, bs, &T_C_L, &k, p->max_integration_distance_m)) continue;
    if (!block_in_band(b, trunc)) continue;
    ensure_layer(b, L_COLOR);
    b->dirty_mesh = 1;
    if (m->n_cview + 1 > m->cview_cap) { m->cview_cap = m->cview_cap ? m->cview_cap * 2 : 1024; m->cview = (Idx3*)realloc(m->cview, (size_t)m->cview_cap * sizeof(Idx3)); }
    m->cview[m->n_cview++] = b->idx;
  }
  const int64_t n = m->n_cview;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < n; i++) {
    Block* b = map_find(m, m->cview[i]);
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      float pl[3] = {voxel_center(b->idx.x, x, bs, vs), voxel_center(b->idx.y, y, bs, vs), voxel_center(b->idx.z, z, bs, vs)};
      float pc[3]; rt_apply(&T_C_L, pl[0], pl[1], pl[2], pc);
      float u, v;
      if (!cam_project(&k, pc, &u, &v)) continue;
      const float vd = pc[2];
      if (p->max_integration_distance_m > 0.0f && vd > p->max_integration_distance_m) continue;
      float sd;
      if (interp_depth(m->synth, m->synth_rows, m->synth_cols, u / (float)f, v / (float)f, 0, &sd) <= 0) continue;
      if (fabsf(sd - vd) > (p->color_occlusion_threshold_vox < 0.0f ? trunc : p->color_occlusion_threshold_vox * vs)) continue;   /* [U] occlusion test */
      float c[3];
      if (!interp_color(rgb, rows, cols, u, v, c)) continue;
      ColorVoxel* cv = &b->color[z + 8 * y + 64 * x];
      const float w0 = cv->weight;
      cv->r = blend_u8((float)cv->r, w0, c[0], 1.0f);
      cv->g = blend_u8((float)cv->g, w0, c[1], 1.0f);
      cv->b = blend_u8((float)cv->b, w0, c[2], 1.0f);
      cv->weight = fminf(w0 + 1.0f, p->max_weight);
    }
  }
  return n;
}
int orc_get_synthetic_depth(const OrcMap* m, float* out, int* rows, int* cols) {
  *rows = m->synth_rows; *cols = m->synth_cols;
  if (out && m->synth) memcpy(out, m->synth, (size_t)m->synth_rows * m->synth_cols * sizeof(float));
  return m->synth != NULL;
}

/* ------------------------------------------------------------------ ESDF (2-D slice) */
typedef struct { int32_t kz_min, kz_max, kz_out, ri; float max_sq, site_dist_m; int plane_on; } EsdfCfg;
/* [U] height of the ground plane n . p + d = 0 at (x, y) (nvbx_internal.h esdf_plane_height, same lines) */
static float esdf_plane_height(const float* pl, float x, float y) {
  float t = pl[0] * x;
  t = t + pl[1] * y;
  t = t + pl[3];
  return -(t / pl[2]);
}
static EsdfCfg esdf_cfg(const OrcParams* p) {
  EsdfCfg c;
  const float vs = p->voxel_size;
  c.kz_min = (int32_t)floorf(p->esdf_slice_min_height / vs);
  c.kz_max = (int32_t)floorf(p->esdf_slice_max_height / vs);
  c.kz_out = (int32_t)floorf(p->esdf_slice_height / vs);
  float r = p->esdf_max_distance_m / vs;
  c.max_sq = r * r;
  c.ri = (int32_t)floorf(r);
  c.site_dist_m = p->esdf_max_site_distance_vox * vs;
  c.plane_on = (p->esdf_use_ground_plane && p->esdf_ground_plane[2] > 1e-3f && p->slice_height_thickness_m > 0.0f) ? 1 : 0;
  return c;
}
/* [U] ground-plane mode: the z band of column (vx, vy) of block column (bx, by), and the block range the band can reach anywhere in that block
 * column (plane heights at its four corners) -- csrc/nvbx_esdf_mark.h esdf_mark_entry, same expressions */
static void esdf_column_band(const OrcParams* p, const EsdfCfg* c, int32_t bx, int32_t by, int vx, int vy, int32_t* kz_lo, int32_t* kz_hi) {
  *kz_lo = c->kz_min; *kz_hi = c->kz_max;
  if (!c->plane_on) return;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float hl = esdf_plane_height(p->esdf_ground_plane, voxel_center(bx, vx, bs, vs), voxel_center(by, vy, bs, vs));
  *kz_lo = (int32_t)floorf((hl + p->slice_height_above_plane_m) / vs);
  *kz_hi = (int32_t)floorf(((hl + p->slice_height_above_plane_m) + p->slice_height_thickness_m) / vs);
}
static void esdf_block_band(const OrcParams* p, const EsdfCfg* c, int32_t bx, int32_t by, int32_t* bz_lo, int32_t* bz_hi) {
  *bz_lo = floor_div8(c->kz_min); *bz_hi = floor_div8(c->kz_max);
  if (!c->plane_on) return;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float x0 = (float)bx * bs, x1 = (float)(bx + 1) * bs, y0 = (float)by * bs, y1 = (float)(by + 1) * bs;
  const float h00 = esdf_plane_height(p->esdf_ground_plane, x0, y0), h10 = esdf_plane_height(p->esdf_ground_plane, x1, y0);
  const float h01 = esdf_plane_height(p->esdf_ground_plane, x0, y1), h11 = esdf_plane_height(p->esdf_ground_plane, x1, y1);
  const float hmin = fminf(fminf(h00, h10), fminf(h01, h11)), hmax = fmaxf(fmaxf(h00, h10), fmaxf(h01, h11));
  *bz_lo = floor_div8((int32_t)floorf((hmin + p->slice_height_above_plane_m) / vs));
  *bz_hi = floor_div8((int32_t)floorf(((hmax + p->slice_height_above_plane_m) + p->slice_height_thickness_m) / vs));
  if (*bz_hi - *bz_lo > 61) *bz_hi = *bz_lo + 61;
}

/* Exact 2-D Euclidean distance transform of the slice with cut-off: row pass (nearest site along x, ties -> -x),
 * column pass (dy ascending, strict improvement).  Restates what [U] EsdfIntegrator's sweep/propagate loop
 * converges to on a convex allocated region; DESIGN.md "ESDF semantics" states the difference elsewhere. */
/* EsdfMode::k3D (node param esdf_mode "3d", node_params.hpp:90): [U] restated as the same thing in three dimensions.  Every
 * dirty TSDF block gets an ESDF block of the same index; every voxel is observed / inside / site by its own TSDF voxel (or
 * occupancy log-odds); a block whose TSDF was deallocated is re-marked (nothing observed).  Distances: exact 3-D Euclidean
 * distance transform with cut-off over all ESDF blocks -- x pass (nearest site along x, ties -> -x), y pass and z pass (minimum
 * of d^2 + previous result over +-ri, ties -> the smaller offset). */
static int64_t update_esdf_3d(OrcMap* m) {
  const OrcParams* p = &m->p;
  const EsdfCfg c = esdf_cfg(p);
  int64_t n_dirty = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    int want = 0;
    if ((b->flags & L_TSDF) && b->dirty_esdf) { b->dirty_esdf = 0; want = 1; }
    if (b->remark_esdf) { b->remark_esdf = 0; if (b->flags & L_ESDF) want = 1; }
    if (!want) continue;
    ensure_layer(b, L_ESDF);
    n_dirty++;
    for (int i = 0; i < NVOX; i++) {
      int observed = 0, inside = 0, site = 0;
      if (b->flags & L_TSDF) {
        const TsdfVoxel* tv = &b->tsdf[i];
        if (p->projective_layer_type == 1) {
          if (tv->distance != 0.0f) observed = 1;
          if (tv->distance > 0.0f) { inside = 1; site = 1; }
        } else if (tv->weight >= p->esdf_min_weight) {
          observed = 1;
          const int in = tv->distance <= 0.0f;
          if (in) inside = 1;
          if ((p->esdf_site_rule == 1 || in) && fabsf(tv->distance) <= c.site_dist_m) site = 1;
        }
      }
      b->esdf[i].observed = (uint8_t)observed; b->esdf[i].is_inside = (uint8_t)inside; b->esdf[i].is_site = (uint8_t)site;
    }
  }
  m->esdf_epoch++;
  int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF)) continue;
    const int32_t v[3] = {b->idx.x, b->idx.y, b->idx.z};
    for (int a = 0; a < 3; a++) { if (v[a] < lo[a]) lo[a] = v[a]; if (v[a] > hi[a]) hi[a] = v[a]; }
  }
  if (lo[0] > hi[0]) return 0;
  const int64_t W = (int64_t)(hi[0] - lo[0] + 1) * 8, H = (int64_t)(hi[1] - lo[1] + 1) * 8, D = (int64_t)(hi[2] - lo[2] + 1) * 8;
  const int64_t N = W * H * D;
  uint8_t* site = (uint8_t*)calloc((size_t)N, 1);
  int16_t* dxs = (int16_t*)malloc((size_t)N * sizeof(int16_t));       /* x pass: offset to the nearest site along x */
  int32_t* sqy = (int32_t*)malloc((size_t)N * sizeof(int32_t));       /* y pass: squared distance in the xy plane */
  int16_t* dys = (int16_t*)malloc((size_t)N * sizeof(int16_t));
  int16_t* dxy = (int16_t*)malloc((size_t)N * sizeof(int16_t));
  const int16_t NONE = 32767;
#define G3(x, y, z) (((int64_t)(z) * H + (y)) * W + (x))
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF)) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++)
      site[G3((int64_t)(b->idx.x - lo[0]) * 8 + x, (int64_t)(b->idx.y - lo[1]) * 8 + y, (int64_t)(b->idx.z - lo[2]) * 8 + z)] = b->esdf[z + 8 * y + 64 * x].is_site;
  }
#pragma omp parallel for
  for (int64_t z = 0; z < D; z++) for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
    int16_t best = NONE;
    for (int32_t d = 0; d <= c.ri; d++) {
      if (x - d >= 0 && site[G3(x - d, y, z)]) { best = (int16_t)(-d); break; }
      if (x + d < W && site[G3(x + d, y, z)]) { best = (int16_t)d; break; }
    }
    dxs[G3(x, y, z)] = best;
  }
#pragma omp parallel for
  for (int64_t z = 0; z < D; z++) for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
    int32_t best = INT32_MAX; int16_t bdy = 0, bdx = 0;
    for (int32_t dy = -c.ri; dy <= c.ri; dy++) {
      const int64_t yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      const int16_t dx = dxs[G3(x, yy, z)];
      if (dx == NONE) continue;
      const int32_t sq = dy * dy + (int32_t)dx * dx;
      if (sq < best) { best = sq; bdy = (int16_t)dy; bdx = dx; }
    }
    sqy[G3(x, y, z)] = best; dys[G3(x, y, z)] = bdy; dxy[G3(x, y, z)] = bdx;
  }
#pragma omp parallel for
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF)) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      const int64_t gx = (int64_t)(b->idx.x - lo[0]) * 8 + x, gy = (int64_t)(b->idx.y - lo[1]) * 8 + y, gz = (int64_t)(b->idx.z - lo[2]) * 8 + z;
      int32_t best = INT32_MAX, bdz = 0; int64_t bi = -1;
      for (int32_t dz = -c.ri; dz <= c.ri; dz++) {
        const int64_t zz = gz + dz;
        if (zz < 0 || zz >= D) continue;
        const int32_t s2 = sqy[G3(gx, gy, zz)];
        if (s2 == INT32_MAX) continue;
        const int32_t sq = dz * dz + s2;
        if (sq < best) { best = sq; bdz = dz; bi = G3(gx, gy, zz); }
      }
      EsdfVoxel* ev = &b->esdf[z + 8 * y + 64 * x];
      if (best != INT32_MAX && (float)best <= c.max_sq) {
        ev->sq = (float)best; ev->parent[0] = dxy[bi]; ev->parent[1] = dys[bi]; ev->parent[2] = bdz;
      } else {
        ev->sq = c.max_sq; ev->parent[0] = 0; ev->parent[1] = 0; ev->parent[2] = 0;
      }
    }
  }
#undef G3
  free(site); free(dxs); free(sqy); free(dys); free(dxy);
  return n_dirty;
}

/* [U] open choice esdf_propagation = 1: the reference's sweep / propagate loop restated as synchronous 4-neighbour parent
 * propagation to its fixed point over the allocated ESDF blocks of the slice (a voxel learns of a site only through a chain
 * of allocated axis neighbours); key = (sq, dy, dx) lexicographic = the exact transform's tie rule; cut-off max_sq; the whole
 * slice is recomputed from its sites on every update.  Dense arrays over the blocks' AABB; `dom` = voxel belongs to a block. */
static void esdf_propagate(OrcMap* m, const EsdfCfg* c, int32_t bz_out, int32_t vz_out) {
  int32_t bx0 = INT32_MAX, bx1 = INT32_MIN, by0 = INT32_MAX, by1 = INT32_MIN;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    if (b->idx.x < bx0) bx0 = b->idx.x; if (b->idx.x > bx1) bx1 = b->idx.x;
    if (b->idx.y < by0) by0 = b->idx.y; if (b->idx.y > by1) by1 = b->idx.y;
  }
  if (bx0 > bx1) return;
  const int64_t W = (int64_t)(bx1 - bx0 + 1) * 8, H = (int64_t)(by1 - by0 + 1) * 8;
  const int32_t NONE = INT32_MAX;
  uint8_t* dom = (uint8_t*)calloc((size_t)(W * H), 1);
  int32_t* cur = (int32_t*)malloc((size_t)(W * H) * sizeof(int32_t));
  int32_t* nxt = (int32_t*)malloc((size_t)(W * H) * sizeof(int32_t));
  for (int64_t i = 0; i < W * H; i++) cur[i] = NONE;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
      const int64_t g = ((int64_t)(b->idx.y - by0) * 8 + y) * W + (int64_t)(b->idx.x - bx0) * 8 + x;
      dom[g] = 1;
      if (b->esdf[vz_out + 8 * y + 64 * x].is_site) cur[g] = (64 << 7) | 64;
    }
  }
  static const int OX[4] = {-1, 1, 0, 0}, OY[4] = {0, 0, -1, 1};
  for (int round = 0; round < (1 << 14); round++) {
    int changed = 0;
#pragma omp parallel for reduction(|:changed)
    for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
      const int64_t g = y * W + x;
      if (!dom[g]) { nxt[g] = NONE; continue; }
      int32_t best = cur[g];
      for (int k = 0; k < 4; k++) {
        const int64_t nx = x + OX[k], ny = y + OY[k];
        if (nx < 0 || ny < 0 || nx >= W || ny >= H || !dom[ny * W + nx]) continue;
        const int32_t nb = cur[ny * W + nx];
        if (nb == NONE) continue;
        const int32_t dx = (nb & 127) - 64 + OX[k], dy = ((nb >> 7) & 127) - 64 + OY[k];
        const int32_t sq = dx * dx + dy * dy;
        if (!((float)sq <= c->max_sq) || dx < -63 || dx > 63 || dy < -63 || dy > 63) continue;
        const int32_t cand = (sq << 14) | ((dy + 64) << 7) | (dx + 64);
        if (cand < best) best = cand;
      }
      nxt[g] = best;
      if (best != cur[g]) changed = 1;
    }
    int32_t* t = cur; cur = nxt; nxt = t;
    if (!changed) break;
  }
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
      const int32_t pk = cur[((int64_t)(b->idx.y - by0) * 8 + y) * W + (int64_t)(b->idx.x - bx0) * 8 + x];
      EsdfVoxel* ev = &b->esdf[vz_out + 8 * y + 64 * x];
      if (pk != NONE) { ev->sq = (float)(pk >> 14); ev->parent[0] = (pk & 127) - 64; ev->parent[1] = ((pk >> 7) & 127) - 64; ev->parent[2] = 0; }
      else { ev->sq = c->max_sq; ev->parent[0] = 0; ev->parent[1] = 0; ev->parent[2] = 0; }
    }
  }
  free(dom); free(cur); free(nxt);
}

int64_t orc_update_esdf(OrcMap* m) {
  if (m->p.esdf_mode == 1) return update_esdf_3d(m);
  const OrcParams* p = &m->p;
  const EsdfCfg c = esdf_cfg(p);
  const int32_t bz_out = floor_div8(c.kz_out), vz_out = mod8(c.kz_out);
  /* 1. allocate + mark sites for dirty TSDF columns intersecting the z band; columns whose TSDF block was
   *    deallocated (decay) are re-marked if their ESDF block exists */
  int64_t n_dirty = 0;
  const int64_t count0 = m->count;
  for (int64_t q = 0; q < count0; q++) {
    Block* tb = m->order[q];
    int want = 0;
    int32_t bz_lo_e, bz_hi_e;                      /* the band's blocks in this block column (fixed, or by the ground plane) */
    esdf_block_band(p, &c, tb->idx.x, tb->idx.y, &bz_lo_e, &bz_hi_e);
    if ((tb->flags & L_TSDF) && tb->dirty_esdf) { tb->dirty_esdf = 0; if (tb->idx.z >= bz_lo_e && tb->idx.z <= bz_hi_e) want = 1; }
    if (tb->remark_esdf) { tb->remark_esdf = 0; want = 2; }
    if (!want) continue;
    Idx3 ei = {tb->idx.x, tb->idx.y, bz_out};
    Block* eb = want == 1 ? map_get_or_create(m, ei) : map_find(m, ei);
    if (!eb) continue;
    if (want == 2 && !(eb->flags & L_ESDF)) continue;
    ensure_layer(eb, L_ESDF);
    if (eb->stamp_esdf == m->esdf_epoch) continue;   /* column already re-marked in this update */
    eb->stamp_esdf = m->esdf_epoch;
    n_dirty++;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
      int observed = 0, inside = 0, site = 0;
      int32_t kz_lo, kz_hi;
      esdf_column_band(p, &c, ei.x, ei.y, x, y, &kz_lo, &kz_hi);
      if (c.plane_on) { if (kz_lo < bz_lo_e * 8) kz_lo = bz_lo_e * 8; if (kz_hi > bz_hi_e * 8 + 7) kz_hi = bz_hi_e * 8 + 7; }      /* (the 62-block cap of the wavefront's probes) */
      for (int32_t kz = kz_lo; kz <= kz_hi; kz++) {
        Idx3 ti = {ei.x, ei.y, floor_div8(kz)};
        Block* b = map_find(m, ti);
        if (!b || !(b->flags & L_TSDF)) continue;
        const TsdfVoxel* tv = &b->tsdf[mod8(kz) + 8 * y + 64 * x];
        if (p->projective_layer_type == 1) {     /* [U] OccupancySiteFunctor: known iff log-odds != 0; site = inside = occupied */
          if (tv->distance != 0.0f) observed = 1;
          if (tv->distance > 0.0f) { inside = 1; site = 1; }
        } else if (tv->weight >= p->esdf_min_weight) {
          observed = 1;
          const int in = tv->distance <= 0.0f;
          if (in) inside = 1;
          if ((p->esdf_site_rule == 1 || in) && fabsf(tv->distance) <= c.site_dist_m) site = 1;
        }
      }
      EsdfVoxel* ev = &eb->esdf[vz_out + 8 * y + 64 * x];
      ev->observed = (uint8_t)observed; ev->is_inside = (uint8_t)inside; ev->is_site = (uint8_t)site;
    }
  }
  m->esdf_epoch++;
  if (p->esdf_propagation == 1) { esdf_propagate(m, &c, bz_out, vz_out); return n_dirty; }
  /* 2. exact EDT over all ESDF blocks of the slice (the HIP path windows this; result identical) */
  int32_t bx0 = INT32_MAX, bx1 = INT32_MIN, by0 = INT32_MAX, by1 = INT32_MIN;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    if (b->idx.x < bx0) bx0 = b->idx.x; if (b->idx.x > bx1) bx1 = b->idx.x;
    if (b->idx.y < by0) by0 = b->idx.y; if (b->idx.y > by1) by1 = b->idx.y;
  }
  if (bx0 > bx1) return 0;
  const int64_t W = (int64_t)(bx1 - bx0 + 1) * 8, H = (int64_t)(by1 - by0 + 1) * 8;
  uint8_t* site = (uint8_t*)calloc((size_t)(W * H), 1);
  int16_t* rowdx = (int16_t*)malloc((size_t)(W * H) * sizeof(int16_t));
  const int16_t NONE = 32767;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++)
      site[((int64_t)(b->idx.y - by0) * 8 + y) * W + (int64_t)(b->idx.x - bx0) * 8 + x] = b->esdf[vz_out + 8 * y + 64 * x].is_site;
  }
#pragma omp parallel for
  for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
    int16_t best = NONE;
    for (int32_t d = 0; d <= c.ri; d++) {
      if (x - d >= 0 && site[y * W + x - d]) { best = (int16_t)(-d); break; }
      if (x + d < W && site[y * W + x + d]) { best = (int16_t)d; break; }
    }
    rowdx[y * W + x] = best;
  }
#pragma omp parallel for
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
      const int64_t gx = (int64_t)(b->idx.x - bx0) * 8 + x, gy = (int64_t)(b->idx.y - by0) * 8 + y;
      int32_t best_sq = INT32_MAX, bdx = 0, bdy = 0;
      for (int32_t dy = -c.ri; dy <= c.ri; dy++) {
        const int64_t yy = gy + dy;
        if (yy < 0 || yy >= H) continue;
        const int16_t dx = rowdx[yy * W + gx];
        if (dx == NONE) continue;
        const int32_t sq = dy * dy + (int32_t)dx * dx;
        if (sq < best_sq) { best_sq = sq; bdx = dx; bdy = dy; }
      }
      EsdfVoxel* ev = &b->esdf[vz_out + 8 * y + 64 * x];
      if (best_sq != INT32_MAX && (float)best_sq <= c.max_sq) {
        ev->sq = (float)best_sq; ev->parent[0] = bdx; ev->parent[1] = bdy; ev->parent[2] = 0;
      } else {
        ev->sq = c.max_sq; ev->parent[0] = 0; ev->parent[1] = 0; ev->parent[2] = 0;
      }
    }
  }
  free(site); free(rowdx);
  return n_dirty;
}

/* [U] EsdfSlicer::sliceLayerToDistanceImage: AABB of allocated ESDF blocks -> rows x cols image, row=y col=x,
 * value = signed metres (esdf_and_gradients_conversions.cu:33-44 functor) or unknown_value.
 * aabb6 = min xyz, max xyz in metres.  Returns rows*cols (0 if no blocks); writes image if cap suffices. */
int64_t orc_esdf_slice_image(const OrcMap* m, float unknown_value, float* img, int64_t cap, int32_t* rows, int32_t* cols, float* aabb6) {
  const OrcParams* p = &m->p;
  const EsdfCfg c = esdf_cfg(p);
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const int32_t bz_out = floor_div8(c.kz_out), vz_out = mod8(c.kz_out);
  int32_t bx0 = INT32_MAX, bx1 = INT32_MIN, by0 = INT32_MAX, by1 = INT32_MIN;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    if (b->idx.x < bx0) bx0 = b->idx.x; if (b->idx.x > bx1) bx1 = b->idx.x;
    if (b->idx.y < by0) by0 = b->idx.y; if (b->idx.y > by1) by1 = b->idx.y;
  }
  if (bx0 > bx1) { *rows = 0; *cols = 0; return 0; }
  const int64_t W = (int64_t)(bx1 - bx0 + 1) * 8, H = (int64_t)(by1 - by0 + 1) * 8;
  *rows = (int32_t)H; *cols = (int32_t)W;
  aabb6[0] = (float)bx0 * bs; aabb6[1] = (float)by0 * bs; aabb6[2] = (float)bz_out * bs;
  aabb6[3] = (float)(bx1 + 1) * bs; aabb6[4] = (float)(by1 + 1) * bs; aabb6[5] = (float)(bz_out + 1) * bs;
  if (W * H > cap) return W * H;
  for (int64_t i = 0; i < W * H; i++) img[i] = unknown_value;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_ESDF) || b->idx.z != bz_out) continue;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
      const EsdfVoxel* ev = &b->esdf[vz_out + 8 * y + 64 * x];
      if (!ev->observed) continue;
      float d = sqrtf(ev->sq) * vs;
      if (ev->is_inside) d = -d;
      img[((int64_t)(b->idx.y - by0) * 8 + y) * W + (int64_t)(b->idx.x - bx0) * 8 + x] = d;
    }
  }
  return W * H;
}

/* voxelLayerToDenseVoxelGridInAABB<SignedDistanceFunctor> restated (esdf_and_gradients_conversions.cu:28-48,88-125):
 * min_vox / size_vox give the AABB in global voxel indices; out[x*(Ny*Nz) + y*Nz + z]. */
void orc_esdf_dense_grid(const OrcMap* m, const int32_t* min_vox, const int32_t* size_vox, float default_value, float* out) {
  const float vs = m->p.voxel_size;
  for (int32_t x = 0; x < size_vox[0]; x++) for (int32_t y = 0; y < size_vox[1]; y++) for (int32_t z = 0; z < size_vox[2]; z++) {
    const int32_t gx = min_vox[0] + x, gy = min_vox[1] + y, gz = min_vox[2] + z;
    Idx3 bi = {floor_div8(gx), floor_div8(gy), floor_div8(gz)};
    const Block* b = map_find(m, bi);
    float v = default_value;
    if (b && (b->flags & L_ESDF)) {
      const EsdfVoxel* ev = &b->esdf[mod8(gz) + 8 * mod8(gy) + 64 * mod8(gx)];
      if (ev->observed) { v = sqrtf(ev->sq) * vs; if (ev->is_inside) v = v * -1.0f; }
    }
    out[((int64_t)x * size_vox[1] + y) * size_vox[2] + z] = v;
  }
}

/* ------------------------------------------------------------------ mesh (marching cubes) */
static const int8_t MC_CORNER[8][3] = {{0,0,0},{1,0,0},{1,1,0},{0,1,0},{0,0,1},{1,0,1},{1,1,1},{0,1,1}};
/* edge -> (lower lattice corner offset, axis) */
static const int8_t MC_EDGE_BASE[12][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,0},{0,0,1},{1,0,1},{0,1,1},{0,0,1},{0,0,0},{1,0,0},{1,1,0},{0,1,0}};
static const int8_t MC_EDGE_AXIS[12] = {0,1,0,1,0,1,0,1,2,2,2,2};

/* [U] MeshIntegrator restated: per block, 8^3 cubes whose 8 corners are the voxel centres (own + +x/+y/+z neighbour
 * blocks); cube skipped if any corner weight < min_weight or neighbour missing; vertices live on lattice edges
 * (lattice corner, axis) so welding = one vertex per crossed edge.  Ordering contract: vertices ascending edge id
 * ((lx*9+ly)*9+lz)*3+axis; triangles in voxel order x-major (x,y,z loops) then table order.  Normal of a welded
 * vertex = normal of the first triangle that references it; colour = colour voxel nearest the vertex (gray 127 if none). */
static int mesh_block(OrcMap* m, Block* b) {
  const OrcParams* p = &m->p;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  static const int L = 9;
  float d[9 * 9 * 9], w[9 * 9 * 9];
  uint8_t has[9 * 9 * 9];
  const ColorVoxel* cptr[9 * 9 * 9];
  Block* nb[8];
  for (int n = 0; n < 8; n++) {
    Idx3 ni = {b->idx.x + (n & 1), b->idx.y + ((n >> 1) & 1), b->idx.z + ((n >> 2) & 1)};
    nb[n] = map_find(m, ni);
    if (nb[n] && !(nb[n]->flags & L_TSDF)) nb[n] = NULL;
  }
  for (int x = 0; x < L; x++) for (int y = 0; y < L; y++) for (int z = 0; z < L; z++) {
    const int n = (x >> 3) | ((y >> 3) << 1) | ((z >> 3) << 2);
    const int li = (x * L + y) * L + z;
    if (!nb[n]) { has[li] = 0; d[li] = 0; w[li] = 0; cptr[li] = NULL; continue; }
    const int vi = (z & 7) + 8 * (y & 7) + 64 * (x & 7);
    has[li] = 1; d[li] = nb[n]->tsdf[vi].distance; w[li] = nb[n]->tsdf[vi].weight;
    cptr[li] = (nb[n]->flags & L_COLOR) ? &nb[n]->color[vi] : NULL;
  }
  int32_t* edge_vid = (int32_t*)malloc(sizeof(int32_t) * 9 * 9 * 9 * 3);
  int32_t* edge_first_tri = (int32_t*)malloc(sizeof(int32_t) * 9 * 9 * 9 * 3);
  for (int i = 0; i < 9 * 9 * 9 * 3; i++) { edge_vid[i] = -1; edge_first_tri[i] = -1; }
  int32_t* tri_edges = (int32_t*)malloc(sizeof(int32_t) * 512 * 5 * 3);
  int32_t ntri = 0;
  for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
    int cube = 0, ok = 1;
    for (int c = 0; c < 8; c++) {
      const int li = ((x + MC_CORNER[c][0]) * L + (y + MC_CORNER[c][1])) * L + (z + MC_CORNER[c][2]);
      if (!has[li] || !(w[li] >= p->mesh_min_weight)) { ok = 0; break; }
      if (d[li] < 0.0f) cube |= 1 << c;
    }
    if (!ok || cube == 0 || cube == 255) continue;
    const int8_t* row = MC_TRI[p->mesh_ambiguity_rule][cube];
    for (int t = 0; row[t] >= 0; t += 3) {
      for (int q = 0; q < 3; q++) {
        const int e = row[t + q];
        const int eid = (((x + MC_EDGE_BASE[e][0]) * L + (y + MC_EDGE_BASE[e][1])) * L + (z + MC_EDGE_BASE[e][2])) * 3 + MC_EDGE_AXIS[e];
        tri_edges[3 * ntri + q] = eid;
        if (edge_first_tri[eid] < 0) edge_first_tri[eid] = ntri;
      }
      ntri++;
    }
  }
  /* vertices in ascending edge id */
  int32_t nvert = 0;
  for (int eid = 0; eid < 9 * 9 * 9 * 3; eid++) if (edge_first_tri[eid] >= 0) edge_vid[eid] = nvert++;
  MeshBlock* mb = &b->mesh;
  free(mb->vert); free(mb->nrm); free(mb->col); free(mb->tri);
  mb->n_vert = nvert; mb->n_tri = ntri;
  mb->vert = (float*)calloc((size_t)(nvert ? nvert : 1) * 3, sizeof(float));
  mb->nrm = (float*)calloc((size_t)(nvert ? nvert : 1) * 3, sizeof(float));
  mb->col = (uint8_t*)calloc((size_t)(nvert ? nvert : 1) * 4, 1);
  mb->tri = (int32_t*)calloc((size_t)(ntri ? ntri : 1) * 3, sizeof(int32_t));
  for (int eid = 0; eid < 9 * 9 * 9 * 3; eid++) {
    const int32_t v = edge_vid[eid];
    if (v < 0) continue;
    const int axis = eid % 3; const int li = eid / 3;
    const int lz = li % 9, ly = (li / 9) % 9, lx = li / 81;
    const int lj = li + (axis == 0 ? 81 : (axis == 1 ? 9 : 1));
    const float da = d[li], db = d[lj];
    const float t = da / (da - db);
    const int32_t l3[3] = {lx, ly, lz};
    const int32_t b3[3] = {b->idx.x, b->idx.y, b->idx.z};
    for (int a = 0; a < 3; a++) {
      float pos = ((float)b3[a] * bs + (float)l3[a] * vs) + vs * 0.5f;
      if (a == axis) pos = pos + t * vs;
      mb->vert[3 * v + a] = pos;
    }
    const ColorVoxel* cv = (t < 0.5f) ? cptr[li] : cptr[lj];
    if (cv && cv->weight > 0.0f) { mb->col[4 * v] = cv->r; mb->col[4 * v + 1] = cv->g; mb->col[4 * v + 2] = cv->b; }
    else { mb->col[4 * v] = 127; mb->col[4 * v + 1] = 127; mb->col[4 * v + 2] = 127; }
    mb->col[4 * v + 3] = 255;
  }
  for (int32_t t = 0; t < ntri; t++) for (int q = 0; q < 3; q++) mb->tri[3 * t + q] = edge_vid[tri_edges[3 * t + q]];
  for (int eid = 0; eid < 9 * 9 * 9 * 3; eid++) {
    const int32_t v = edge_vid[eid];
    if (v < 0) continue;
    /* normal rule 0: the first triangle referencing the vertex; rule 1: sum of the unnormalised normals (= area-weighted mean) of
     * every triangle of the block referencing it, ascending triangle index */
    float n[3] = {0.0f, 0.0f, 0.0f};
    const int32_t t_lo = p->mesh_normal_rule == 0 ? edge_first_tri[eid] : 0, t_hi = p->mesh_normal_rule == 0 ? edge_first_tri[eid] + 1 : ntri;
    for (int32_t t = t_lo; t < t_hi; t++) {
      if (p->mesh_normal_rule != 0 && mb->tri[3 * t] != v && mb->tri[3 * t + 1] != v && mb->tri[3 * t + 2] != v) continue;
      const float* p0 = &mb->vert[3 * mb->tri[3 * t]]; const float* p1 = &mb->vert[3 * mb->tri[3 * t + 1]]; const float* p2 = &mb->vert[3 * mb->tri[3 * t + 2]];
      const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      n[0] = n[0] + (e1[1] * e2[2] - e1[2] * e2[1]); n[1] = n[1] + (e1[2] * e2[0] - e1[0] * e2[2]); n[2] = n[2] + (e1[0] * e2[1] - e1[1] * e2[0]);
    }
    const float len = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    if (len > 0.0f) { n[0] = n[0] / len; n[1] = n[1] / len; n[2] = n[2] / len; }
    mb->nrm[3 * v] = n[0]; mb->nrm[3 * v + 1] = n[1]; mb->nrm[3 * v + 2] = n[2];
  }
  free(edge_vid); free(edge_first_tri); free(tri_edges);
  b->flags |= L_MESH;
  return ntri;
}

/* Mapper::updateColorMesh(UpdateFullLayer) restated at call-site level (layer_publishing.cpp:686-689). Returns #blocks meshed. */
int64_t orc_update_mesh(OrcMap* m, int full) {
  int64_t n = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_TSDF)) continue;
    if (!full && !b->dirty_mesh) continue;
    b->dirty_mesh = 0;
    mesh_block(m, b);
    n++;
  }
  return n;
}
int orc_mesh_counts(const OrcMap* m, int32_t x, int32_t y, int32_t z, int32_t* n_vert, int32_t* n_tri) {
  Idx3 i = {x, y, z}; Block* b = map_find(m, i);
  if (!b || !(b->flags & L_MESH)) return 0;
  *n_vert = b->mesh.n_vert; *n_tri = b->mesh.n_tri; return 1;
}
int orc_mesh_get(const OrcMap* m, int32_t x, int32_t y, int32_t z, float* vert, float* nrm, uint8_t* col, int32_t* tri) {
  Idx3 i = {x, y, z}; Block* b = map_find(m, i);
  if (!b || !(b->flags & L_MESH)) return 0;
  memcpy(vert, b->mesh.vert, (size_t)b->mesh.n_vert * 12); memcpy(nrm, b->mesh.nrm, (size_t)b->mesh.n_vert * 12);
  memcpy(col, b->mesh.col, (size_t)b->mesh.n_vert * 4); memcpy(tri, b->mesh.tri, (size_t)b->mesh.n_tri * 12);
  return 1;
}

/* ------------------------------------------------------------------ ground plane ([U] GroundPlaneEstimator restated; nvblox_node.cpp:1456,1474)
 * candidates = upward zero crossings of the TSDF (d(z) <= 0 < d(z + 1), both observed, linear interpolation along z) with height in
 * [min_z, max_z]; plane = RANSAC over them with a fixed linear congruential sampling sequence */
static int xyz_cmp(const void* a, const void* b) {
  const float* p = (const float*)a; const float* q = (const float*)b;
  for (int i = 0; i < 3; i++) if (p[i] != q[i]) return p[i] < q[i] ? -1 : 1;
  return 0;
}
int64_t orc_tsdf_zero_crossings(OrcMap* m, float min_z, float max_z, float* out_xyz, int64_t cap) {
  const OrcParams* p = &m->p;
  const float vs = p->voxel_size, bs = vs * 8.0f;
  const float min_w = p->esdf_min_weight > 0.0f ? p->esdf_min_weight : 1e-4f;
  int64_t n = 0;
  for (int64_t q = 0; q < m->count; q++) {
    const Block* b = m->order[q];
    if (!(b->flags & L_TSDF)) continue;
    Idx3 ui = {b->idx.x, b->idx.y, b->idx.z + 1};
    const Block* up = map_find(m, ui);
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      const TsdfVoxel lo = b->tsdf[z + 8 * y + 64 * x];
      TsdfVoxel hi = {0.0f, 0.0f};
      if (z < 7) hi = b->tsdf[z + 1 + 8 * y + 64 * x];
      else if (up && (up->flags & L_TSDF)) hi = up->tsdf[0 + 8 * y + 64 * x];
      if (!(lo.weight >= min_w && hi.weight >= min_w && lo.distance <= 0.0f && hi.distance > 0.0f)) continue;
      const float z_lo = ((float)b->idx.z * bs + (float)z * vs) + vs * 0.5f;
      const float pz = z_lo + vs * (-lo.distance / (hi.distance - lo.distance));
      if (!(pz >= min_z && pz <= max_z)) continue;
      if (n < cap) { out_xyz[3 * n] = ((float)b->idx.x * bs + (float)x * vs) + vs * 0.5f; out_xyz[3 * n + 1] = ((float)b->idx.y * bs + (float)y * vs) + vs * 0.5f; out_xyz[3 * n + 2] = pz; }
      n++;
    }
  }
  if (n <= cap) qsort(out_xyz, (size_t)n, 3 * sizeof(float), xyz_cmp);
  return n;
}
int64_t orc_fit_plane_ransac(const float* pts, int64_t n, float thresh, int32_t iterations, uint32_t seed, float* plane) {
  plane[0] = 0.0f; plane[1] = 0.0f; plane[2] = 1.0f; plane[3] = 0.0f;
  if (n < 3) return 0;
  uint32_t state = seed;
  const uint32_t mod = (uint32_t)(n < (1 << 24) ? n : (1 << 24));
  int64_t best = 0;
  for (int32_t it = 0; it < iterations; it++) {
    int64_t id[3];
    for (int k = 0; k < 3; k++) { state = state * 1664525u + 1013904223u; id[k] = (int64_t)((state >> 8) % mod); }
    if (id[0] == id[1] || id[0] == id[2] || id[1] == id[2]) continue;
    const float* a = pts + 3 * id[0]; const float* b = pts + 3 * id[1]; const float* c = pts + 3 * id[2];
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    if (!(len > 1e-12f)) continue;
    nx = nx / len; ny = ny / len; nz = nz / len;
    if (nz < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    const float d = -((nx * a[0] + ny * a[1]) + nz * a[2]);
    int64_t inl = 0;
    for (int64_t k = 0; k < n; k++) { const float* q = pts + 3 * k; if (fabsf(((nx * q[0] + ny * q[1]) + nz * q[2]) + d) <= thresh) inl++; }
    if (inl > best) { best = inl; plane[0] = nx; plane[1] = ny; plane[2] = nz; plane[3] = d; }
  }
  return best;
}

/* ------------------------------------------------------------------ decay / clearing */
static void cleared_push(OrcMap* m, Idx3 i) {
  if (m->n_cleared + 1 > m->cleared_cap) { m->cleared_cap = m->cleared_cap ? m->cleared_cap * 2 : 1024; m->cleared = (Idx3*)realloc(m->cleared, (size_t)m->cleared_cap * sizeof(Idx3)); }
  m->cleared[m->n_cleared++] = i;
}
/* Mapper::decayTsdf restated (nvblox_node.cpp:931-936; params nvblox_base.yaml:103-107): weight *= factor for every
 * TSDF voxel of every block NOT in the last depth view; block deallocated when all weights < threshold. */
int64_t orc_decay_tsdf(OrcMap* m, int exclude_last_view) {
  const OrcParams* p = &m->p;
  int64_t removed = 0, keep = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    int drop = 0;
    if ((b->flags & L_TSDF) && !(exclude_last_view && m->camera_frame > 0 && b->stamp_cam == m->camera_frame)) {
      int alive = 0;
      /* [U] switches: tsdf_set_free_distance_on_decayed -- an OBSERVED voxel whose weight falls below the threshold becomes free (distance
       * = tsdf_decayed_free_distance_vox voxels, weight = the threshold) instead of fading to unknown; decay_integrator_deallocate_decayed_blocks
       * = false -- a fully decayed block stays allocated with its decayed voxels */
      const float free_dist = p->tsdf_decayed_free_distance_vox * p->voxel_size;
      for (int i = 0; i < NVOX; i++) {
        const float w0 = b->tsdf[i].weight;
        float w = w0 * p->tsdf_decay_factor;
        if (!(w < p->tsdf_decayed_weight_threshold)) alive = 1;
        else if (p->tsdf_set_free_distance_on_decayed && w0 > 0.0f) { b->tsdf[i].distance = free_dist; w = p->tsdf_decayed_weight_threshold; }
        b->tsdf[i].weight = w;
      }
      if (!p->decay_deallocate_decayed_blocks) alive = 1;
      b->dirty_esdf = 1; b->dirty_mesh = 1;
      if (!alive) {
        drop = 1; cleared_push(m, b->idx);
        const EsdfCfg ec = esdf_cfg(p);
        if (p->esdf_mode == 1) { if (b->flags & L_ESDF) b->remark_esdf = 1; }      /* 3-D: the block's own ESDF block */
        else if (ec.plane_on || (b->idx.z >= floor_div8(ec.kz_min) && b->idx.z <= floor_div8(ec.kz_max))) {      /* (ground-plane mode: a block of any height) */
          Idx3 ei = {b->idx.x, b->idx.y, floor_div8(ec.kz_out)};
          Block* eb = map_find(m, ei);
          if (eb && (eb->flags & L_ESDF)) eb->remark_esdf = 1;
        }
      }
    }
    if (drop && !(b->flags & L_ESDF)) { block_free(b); removed++; }
    else {
      if (drop) { free(b->tsdf); b->tsdf = NULL; free(b->color); b->color = NULL; free(b->fs); b->fs = NULL; b->flags &= ~(uint32_t)(L_TSDF | L_COLOR | L_MESH | L_FREESPACE); removed++; }
      m->order[keep++] = b;
    }
  }
  m->count = keep;
  if (removed) map_rebuild(m);
  return removed;
}
/* [U] Mapper::decayOccupancyAllVoxels restated (nvblox_node.cpp:925-929): log-odds move towards 0 by the log-odds of the
 * region's decay probability and stop there; all-unknown blocks are deallocated (ESDF column re-marked like decayTsdf). */
int64_t orc_decay_occupancy(OrcMap* m) {
  const OrcParams* p = &m->p;
  const float lo_free = log_odds(p->free_region_decay_probability), lo_occ = log_odds(p->occupied_region_decay_probability);
  int64_t removed = 0, keep = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    int drop = 0;
    if (b->flags & L_TSDF) {
      int alive = 0;
      for (int i = 0; i < NVOX; i++) {
        float v = b->tsdf[i].distance;
        /* [U] occupancy_decay_to_free: occupied voxels decay past unknown into free and stay there; free voxels are not decayed */
        if (v > 0.0f) { v = v + lo_occ; if (v < 0.0f && !p->occupancy_decay_to_free) v = 0.0f; }
        else if (v < 0.0f && !p->occupancy_decay_to_free) { v = v + lo_free; if (v > 0.0f) v = 0.0f; }
        b->tsdf[i].distance = v; b->tsdf[i].weight = 0.0f;
        if (v != 0.0f) alive = 1;
      }
      if (!p->decay_deallocate_decayed_blocks) alive = 1;
      if (alive) b->dirty_esdf = 1;
      else {
        drop = 1; cleared_push(m, b->idx);
        const EsdfCfg ec = esdf_cfg(p);
        if (p->esdf_mode == 1) { if (b->flags & L_ESDF) b->remark_esdf = 1; }      /* 3-D: the block's own ESDF block */
        else if (ec.plane_on || (b->idx.z >= floor_div8(ec.kz_min) && b->idx.z <= floor_div8(ec.kz_max))) {      /* (ground-plane mode: a block of any height) */
          Idx3 ei = {b->idx.x, b->idx.y, floor_div8(ec.kz_out)};
          Block* eb = map_find(m, ei);
          if (eb && (eb->flags & L_ESDF)) eb->remark_esdf = 1;
        }
      }
    }
    if (drop && !(b->flags & L_ESDF)) { block_free(b); removed++; }
    else {
      if (drop) { free(b->tsdf); b->tsdf = NULL; free(b->color); b->color = NULL; free(b->fs); b->fs = NULL; b->flags &= ~(uint32_t)(L_TSDF | L_COLOR | L_MESH | L_FREESPACE); b->dirty_esdf = 0; removed++; }
      m->order[keep++] = b;
    }
  }
  m->count = keep;
  if (removed) map_rebuild(m);
  return removed;
}
/* Mapper::clearOutsideRadius restated (nvblox_node.cpp:1566-1583): drop blocks whose centre is farther than r from c. */
int64_t orc_clear_outside_radius(OrcMap* m, const float* c, float r) {
  const float bs = m->p.voxel_size * 8.0f;
  int64_t removed = 0, keep = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    float dx = ((float)b->idx.x * bs + bs * 0.5f) - c[0], dy = ((float)b->idx.y * bs + bs * 0.5f) - c[1], dz = ((float)b->idx.z * bs + bs * 0.5f) - c[2];
    float d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 > r * r) { if (b->flags & L_TSDF) cleared_push(m, b->idx); block_free(b); removed++; } else m->order[keep++] = b;
  }
  m->count = keep;
  if (removed) map_rebuild(m);
  return removed;
}

/* Mapper::getClearedBlocks restated at call-site level (layer_publishing.cpp:716): sorted unique indices, list emptied */
static int idx_cmp(const void* a, const void* b);
int64_t orc_take_cleared_blocks(OrcMap* m, int32_t* out, int64_t cap) {
  qsort(m->cleared, (size_t)m->n_cleared, sizeof(Idx3), idx_cmp);
  int64_t u = 0;
  for (int64_t i = 0; i < m->n_cleared; i++) {
    if (i > 0 && idx_cmp(&m->cleared[i], &m->cleared[i - 1]) == 0) continue;
    if (u < cap) { out[3 * u] = m->cleared[i].x; out[3 * u + 1] = m->cleared[i].y; out[3 * u + 2] = m->cleared[i].z; }
    u++;
  }
  m->n_cleared = 0;
  return u;
}

/* Mapper::clearTsdfInsideShapes restated (nvblox_node.cpp:1834): voxels whose centre is inside a sphere / box are reset.
 * shapes: n x 7 floats {kind, a0, a1, a2, b0, b1, b2}. */
int64_t orc_clear_tsdf_inside_shapes(OrcMap* m, const float* shapes, int32_t n) {
  const float vs = m->p.voxel_size, bs = vs * 8.0f;
  int64_t cleared = 0;
  for (int64_t q = 0; q < m->count; q++) {
    Block* b = m->order[q];
    if (!(b->flags & L_TSDF)) continue;
    int touched = 0;
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) for (int z = 0; z < 8; z++) {
      const float px = voxel_center(b->idx.x, x, bs, vs), py = voxel_center(b->idx.y, y, bs, vs), pz = voxel_center(b->idx.z, z, bs, vs);
      int inside = 0;
      for (int32_t k = 0; k < n && !inside; k++) {
        const float* s7 = shapes + 7 * k;
        if (s7[0] == 0.0f) {
          const float dx = px - s7[1], dy = py - s7[2], dz = pz - s7[3];
          inside = ((dx * dx + dy * dy) + dz * dz) <= s7[4] * s7[4];
        } else {
          inside = px >= s7[1] && py >= s7[2] && pz >= s7[3] && px <= s7[4] && py <= s7[5] && pz <= s7[6];
        }
      }
      if (inside) { TsdfVoxel* v = &b->tsdf[z + 8 * y + 64 * x]; v->distance = 0.0f; v->weight = 0.0f; touched = 1; cleared++; }
    }
    if (touched) { b->dirty_esdf = 1; b->dirty_mesh = 1; }
  }
  return cleared;
}

/* [U] LiDAR motion compensation restated (nvbx_motion_math.h holds the shared per-point arithmetic) */
void orc_motion_compensate_pointcloud(const float* in, const float* rel_ms, int64_t n, const float* T0, const float* T1, float duration_ms, float* out) {
  const nvbx_rel_motion mo = nvbx_rel_motion_make(T0, T1);
  const float inv = 1.0f / duration_ms;
  for (int64_t i = 0; i < n; i++) {
    float a = rel_ms[i] * inv;
    if (!(a > 0.0f)) a = 0.0f;
    if (a > 1.0f) a = 1.0f;
    nvbx_motion_compensate_point(&mo, a, in + 3 * i, out + 3 * i);
  }
}

/* ------------------------------------------------------------------ mask splitting (human mapping) */
/* [U] ImageMasker::splitImageOnGPU restated (MultiMapper::integrateDepth with a mask, nvblox_node.cpp:1018-1060): a valid depth
 * pixel is lifted, moved into the mask camera by T_CM_CD and projected; masked iff it lands on a non-zero mask pixel and its
 * depth there is within `thr` of the nearest depth pixel landing on the same mask pixel.  Invalid value -1. */
static int32_t mask_pixel(const Rt* T, const Cam* dc, const Cam* mc, int32_t mrows, int32_t mcols, int32_t r, int32_t c, float d, float* z_cm) {
  const float rx = (((float)c + 0.5f) - dc->cu) / dc->fu, ry = (((float)r + 0.5f) - dc->cv) / dc->fv;
  float p[3]; rt_apply(T, d * rx, d * ry, d, p);
  *z_cm = p[2];
  if (p[2] <= 0.0f) return -1;
  const float u = mc->fu * (p[0] / p[2]) + mc->cu, v = mc->fv * (p[1] / p[2]) + mc->cv;
  const int32_t cc = (int32_t)floorf(u), rr = (int32_t)floorf(v);
  if (cc < 0 || rr < 0 || cc >= mcols || rr >= mrows) return -1;
  return rr * mcols + cc;
}
void orc_split_depth_by_mask(const float* depth, int32_t rows, int32_t cols, const uint8_t* mask, int32_t mrows, int32_t mcols,
                             const float* T_CM_CD, const float* dcam6, const float* mcam6, float thr, float* unmasked, float* masked) {
  Rt T, Tinv; rt_from_T(T_CM_CD, &T, &Tinv);
  const Cam dc = cam_from(dcam6), mc = cam_from(mcam6);
  float* zmin = (float*)malloc(sizeof(float) * (size_t)mrows * mcols);
  for (int64_t i = 0; i < (int64_t)mrows * mcols; i++) zmin[i] = 3.0e38f;
  for (int32_t r = 0; r < rows; r++) for (int32_t c = 0; c < cols; c++) {
    const float d = depth[(int64_t)r * cols + c];
    if (!(d > 0.0f)) continue;
    float z; const int32_t mi = mask_pixel(&T, &dc, &mc, mrows, mcols, r, c, d, &z);
    if (mi >= 0 && z < zmin[mi]) zmin[mi] = z;
  }
  for (int32_t r = 0; r < rows; r++) for (int32_t c = 0; c < cols; c++) {
    const int64_t i = (int64_t)r * cols + c;
    const float d = depth[i];
    int is_masked = 0;
    if (d > 0.0f) {
      float z; const int32_t mi = mask_pixel(&T, &dc, &mc, mrows, mcols, r, c, d, &z);
      if (mi >= 0 && mask[mi] != 0 && z <= zmin[mi] + thr) is_masked = 1;
    }
    unmasked[i] = (d > 0.0f && is_masked) ? -1.0f : d;
    masked[i] = (d > 0.0f && is_masked) ? d : -1.0f;
  }
  free(zmin);
}

/* multi-GPU union step: TSDF blocks another mapper updated become ESDF-dirty here if they exist locally */
int64_t orc_mark_esdf_dirty(OrcMap* m, const int32_t* idx, int64_t n) {
  int64_t hit = 0;
  for (int64_t i = 0; i < n; i++) {
    Idx3 k = {idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]};
    Block* b = map_find(m, k);
    if (b && (b->flags & L_TSDF)) { b->dirty_esdf = 1; hit++; }
  }
  return hit;
}
/* Index3D of the TSDF blocks currently ESDF-dirty (sorted) */
int64_t orc_esdf_dirty_list(const OrcMap* m, int32_t* out, int64_t cap) {
  int64_t n = 0;
  for (int64_t k = 0; k < m->count; k++) if ((m->order[k]->flags & L_TSDF) && m->order[k]->dirty_esdf) {
    if (n < cap) { out[3 * n] = m->order[k]->idx.x; out[3 * n + 1] = m->order[k]->idx.y; out[3 * n + 2] = m->order[k]->idx.z; }
    n++;
  }
  qsort(out, (size_t)(n < cap ? n : cap), sizeof(Idx3), idx_cmp);
  return n;
}
