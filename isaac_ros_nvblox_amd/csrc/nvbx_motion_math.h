/* nvbx_motion_math.h -- plain-C pose interpolation for LiDAR motion compensation, shared by the HIP kernel (convert.hip) and the
 * oracle so that both sides produce identical bits (multiplications, additions, one square root, divisions: no libm trig on the
 * per-point path).  [U] restated: MultiMapper::integrateDepth(pointcloud, T_L_S, lidar, use_lidar_motion_compensation,
 * T_L_S_scanEnd, scan_duration_ms, ...) -- nvblox_node.cpp:1339-1384: every point was measured at its own time within the scan;
 * the sensor pose at that time is interpolated between scan start and scan end (translation linearly, rotation by normalised
 * quaternion interpolation -- the rotation over a 0.1 s scan is a few degrees) and the point is re-expressed in the sensor
 * frame at scan START, the frame T_L_S refers to. */
#ifndef NVBX_MOTION_MATH_H_
#define NVBX_MOTION_MATH_H_
#ifdef __HIPCC__
#define NVBX_MM_FN __host__ __device__ static inline
#else
#include <math.h>
#define NVBX_MM_FN static inline
#endif

typedef struct { float qw, qx, qy, qz; float t[3]; } nvbx_rel_motion;     /* sensor at scan END expressed in the sensor at scan START */

/* T0, T1: row-major 4x4 rigid transforms T_L_S at scan start / end.  Host side (uses sqrtf only). */
NVBX_MM_FN nvbx_rel_motion nvbx_rel_motion_make(const float* T0, const float* T1) {
  float R[9], t[3];                                   /* R = R0^T R1, t = R0^T (t1 - t0) */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    float s = T0[4 * 0 + i] * T1[4 * 0 + j];
    s = s + T0[4 * 1 + i] * T1[4 * 1 + j];
    s = s + T0[4 * 2 + i] * T1[4 * 2 + j];
    R[3 * i + j] = s;
  }
  const float d[3] = {T1[3] - T0[3], T1[7] - T0[7], T1[11] - T0[11]};
  for (int i = 0; i < 3; i++) { float s = T0[4 * 0 + i] * d[0]; s = s + T0[4 * 1 + i] * d[1]; s = s + T0[4 * 2 + i] * d[2]; t[i] = s; }
  nvbx_rel_motion m;
  /* quaternion of R (w >= 0 branch: the rotation within one scan is far below 180 degrees; general branch kept for safety) */
  const float tr = (R[0] + R[4]) + R[8];
  if (tr > 0.0f) {
    const float s = sqrtf(tr + 1.0f) * 2.0f;
    m.qw = 0.25f * s; m.qx = (R[7] - R[5]) / s; m.qy = (R[2] - R[6]) / s; m.qz = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const float s = sqrtf(((1.0f + R[0]) - R[4]) - R[8]) * 2.0f;
    m.qw = (R[7] - R[5]) / s; m.qx = 0.25f * s; m.qy = (R[1] + R[3]) / s; m.qz = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const float s = sqrtf(((1.0f + R[4]) - R[0]) - R[8]) * 2.0f;
    m.qw = (R[2] - R[6]) / s; m.qx = (R[1] + R[3]) / s; m.qy = 0.25f * s; m.qz = (R[5] + R[7]) / s;
  } else {
    const float s = sqrtf(((1.0f + R[8]) - R[0]) - R[4]) * 2.0f;
    m.qw = (R[3] - R[1]) / s; m.qx = (R[2] + R[6]) / s; m.qy = (R[5] + R[7]) / s; m.qz = 0.25f * s;
  }
  if (m.qw < 0.0f) { m.qw = -m.qw; m.qx = -m.qx; m.qy = -m.qy; m.qz = -m.qz; }      /* shortest path */
  m.t[0] = t[0]; m.t[1] = t[1]; m.t[2] = t[2];
  return m;
}
/* point p measured at fraction a in [0, 1] of the scan -> the same point in the sensor frame at scan start */
NVBX_MM_FN void nvbx_motion_compensate_point(const nvbx_rel_motion* m, float a, const float* p, float* o) {
  float w = (1.0f - a) + a * m->qw, x = a * m->qx, y = a * m->qy, z = a * m->qz;
  const float n = sqrtf(((w * w + x * x) + y * y) + z * z);
  w = w / n; x = x / n; y = y / n; z = z / n;
  /* p' = p + 2 w (v x p) + 2 v x (v x p) */
  const float cx = y * p[2] - z * p[1], cy = z * p[0] - x * p[2], cz = x * p[1] - y * p[0];
  const float dx = y * cz - z * cy, dy = z * cx - x * cz, dz = x * cy - y * cx;
  o[0] = (p[0] + 2.0f * (w * cx)) + 2.0f * dx + a * m->t[0];
  o[1] = (p[1] + 2.0f * (w * cy)) + 2.0f * dy + a * m->t[1];
  o[2] = (p[2] + 2.0f * (w * cz)) + 2.0f * dz + a * m->t[2];
}
#endif
