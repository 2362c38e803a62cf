/* nvbx_lidar_math.h -- spinning-LiDAR sensor model shared by the HIP kernels and (for bit-parity of the projection) by
 * the CPU oracle.  Plain C; every operation is an IEEE add / mul / fma / div / sqrt in a fixed order (both sides build with
 * -ffp-contract=off; fused multiply-adds are written out, nvbx_arith.h), so host and device produce identical bits -- libm's atan2f/acosf and the device's ocml versions
 * differ in the last ulp, which would flip pixel taps and block boundaries between the two.
 *
 * Model ([U] nvblox sensors/lidar restated; anchors in the reference: Lidar(w, h, min_range, vfov) and
 * Lidar(w, h, min_range, min_below, max_above) nvblox_ros/src/lib/nvblox_node.cpp:1315-1323; "each point projects to a
 * pixel center" conversions/pointcloud_conversions.cu:78-92; elevation = asin(z/r), azimuth = atan2(y, x), equal angular
 * bins nvblox_ros/scripts/calculate_lidar_params.py:50-58):
 *   rows = elevation divisions (row 0 = highest beam), cols = azimuth divisions over 360 deg;
 *   beam (k, j) passes through pixel CENTRE (j + 0.5, k + 0.5): elevation = max_el - k * rpp_el, azimuth = -pi + j * rpp_az,
 *   rpp_el = (max_el - min_el) / (rows - 1), rpp_az = 2 pi / cols;   depth of a point = its range |p|.
 */
#ifndef NVBX_LIDAR_MATH_H_
#define NVBX_LIDAR_MATH_H_
#include <math.h>
#include <stdint.h>
#include "nvbx_arith.h"

#if defined(__HIPCC__)
#define NVBX_HD __host__ __device__ inline
#else
#define NVBX_HD static inline
#endif

#define NVBX_PI_F 3.14159274101257324f      /* float(pi) */
#define NVBX_HALF_PI_F 1.57079637050628662f
#define NVBX_QUARTER_PI_F 0.785398185253143311f

/* atan(num / den) for 0 <= num <= den, den > 0: Cephes atanf scheme (reduction at tan(pi/8), degree-4 polynomial in x^2),
 * |err| < 2e-7.  ONE division: the reduced argument (t - 1) / (t + 1) with t = num / den is (num - den) / (num + den), so the
 * operands are selected first and divided once; the polynomial is a Horner chain of fused multiply-adds (the LiDAR integrator
 * evaluates this per voxel and is VALU-bound). */
NVBX_HD float nvbx_atan_ratio(float num, float den) {
  float y0 = 0.0f, n = num, d = den;
  if (num > 0.414213568f * den) { y0 = NVBX_QUARTER_PI_F; n = num - den; d = num + den; }
  const float x = NVBX_DIV(n, d);
  const float z = x * x;
  float p = NVBX_FMA(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = NVBX_FMA(p, z, 1.99777106478e-1f);
  p = NVBX_FMA(p, z, -3.33329491539e-1f);
  p = p * z;
  p = NVBX_FMA(p, x, x);
  return y0 + p;
}
NVBX_HD float nvbx_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  if (ax == 0.0f && ay == 0.0f) return 0.0f;
  /* one polynomial for both octants (operands selected first): (ax >= ay) ? atan(ay / ax) : pi/2 - atan(ax / ay) */
  const float num = (ax >= ay) ? ay : ax, den = (ax >= ay) ? ax : ay;
  float a = nvbx_atan_ratio(num, den);
  if (!(ax >= ay)) a = NVBX_HALF_PI_F - a;
  if (x < 0.0f) a = NVBX_PI_F - a;
  return y < 0.0f ? -a : a;
}
/* asin(s) for |s| <= 0.5 (elevations up to 30 degrees: every beam of a spinning LiDAR): s + s^3 P(s^2), P = degree-4 minimax fit
 * on [0, 0.51] (tools/fit_asin.py), |err| < 4e-8 in float.  No square root, no reduction. */
NVBX_HD float nvbx_asin_small(float s) {
  const float z = s * s;
  float p = NVBX_FMA(4.511628299951553e-2f, z, 2.232578955590725e-2f);
  p = NVBX_FMA(p, z, 4.5879658311605453e-2f);
  p = NVBX_FMA(p, z, 7.491616904735565e-2f);
  p = NVBX_FMA(p, z, 1.666686236858368e-1f);
  p = p * z;
  return NVBX_FMA(p, s, s);
}

typedef struct {
  int32_t cols, rows;             /* azimuth divisions, elevation divisions */
  float min_valid_range_m;
  float max_el, rpp_el, rpp_az;   /* derived: highest beam elevation, radians per pixel */
  float ppr_el, ppr_az;           /* derived: pixels per radian (1 / rpp, rounded once) */
} nvbx_lidar_model;

NVBX_HD nvbx_lidar_model nvbx_lidar_make(int32_t cols, int32_t rows, float min_range, float min_el, float max_el) {
  nvbx_lidar_model l;
  l.cols = cols; l.rows = rows; l.min_valid_range_m = min_range; l.max_el = max_el;
  l.rpp_el = (max_el - min_el) / (float)(rows - 1);
  l.rpp_az = (2.0f * NVBX_PI_F) / (float)cols;
  l.ppr_el = 1.0f / l.rpp_el; l.ppr_az = 1.0f / l.rpp_az;
  return l;
}
NVBX_HD float nvbx_lidar_range(const float* p) { return NVBX_SQRT((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]); }

/* Lidar::project: p (sensor frame), r = its range -> corner-referenced image coordinates (u along azimuth, v along elevation).
 * Elevation = asin(z / r): the range is needed anyway (it is the voxel's depth), so the common case is a short polynomial instead
 * of a second square root (rho), an octant selection and the atan reduction; steeper than 30 degrees (never a beam of the sensors
 * this models, but a voxel next to the sensor can be) falls back to atan2(z, rho).
 * ONE reciprocal serves both angles: with (n, d) the azimuth's reduced operands (nvbx_atan_ratio's selection), q = 1 / (r d) gives
 * sin(elevation) = (z d) q and the reduced tangent n / d = (n r) q -- four multiplications for the second IEEE division. */
NVBX_HD int nvbx_lidar_project(const nvbx_lidar_model* l, const float* p, float r, float* u, float* v) {
  if (r < l->min_valid_range_m || !(r > 1.0e-9f) || !(r < 1.0e18f)) return 0;      /* (a point within a nanometre of the sensor has no direction; keeps r d inside NVBX_DIV's range) */
  const float ax = fabsf(p[0]), ay = fabsf(p[1]);
  const int hi = !(ax >= ay);                                   /* second octant: azimuth = pi/2 - atan(ax / ay) */
  const float num = hi ? ax : ay, den = hi ? ay : ax;           /* 0 <= num <= den */
  float y0 = 0.0f, n = num, d = den;
  if (num > 0.414213568f * den) { y0 = NVBX_QUARTER_PI_F; n = num - den; d = num + den; }
  if (den == 0.0f) d = 1.0f;                                    /* on the sensor's z axis: n = 0, azimuth 0 by convention */
  const float q = NVBX_DIV(1.0f, r * d);
  const float s = (p[2] * d) * q;
  float el = nvbx_asin_small(s);
  if (NVBX_ANY_LANE(fabsf(s) > 0.5f)) {
    float rho2 = p[0] * p[0] + p[1] * p[1];
    NVBX_KEEP_HERE(rho2);
    if (fabsf(s) > 0.5f) el = nvbx_atan2f(p[2], NVBX_SQRT(rho2));
  }
  const float vv = NVBX_FMA(l->max_el - el, l->ppr_el, 0.5f);
  if (!(vv >= 0.0f && vv < (float)l->rows)) return 0;           /* outside the vertical field of view (written so that a NaN is outside) */
  const float t = (n * r) * q;
  const float z = t * t;
  float c = NVBX_FMA(8.05374449538e-2f, z, -1.38776856032e-1f);        /* nvbx_atan_ratio's polynomial */
  c = NVBX_FMA(c, z, 1.99777106478e-1f);
  c = NVBX_FMA(c, z, -3.33329491539e-1f);
  c = c * z;
  c = NVBX_FMA(c, t, t);
  float az = y0 + c;
  if (hi) az = NVBX_HALF_PI_F - az;
  if (p[0] < 0.0f) az = NVBX_PI_F - az;
  if (p[1] < 0.0f) az = -az;
  float uu = NVBX_FMA(az + NVBX_PI_F, l->ppr_az, 0.5f);
  if (uu >= (float)l->cols) uu = uu - (float)l->cols;           /* azimuth wrap-around */
  if (!(uu >= 0.0f && uu < (float)l->cols)) return 0;
  *u = uu; *v = vv;
  return 1;
}
#endif
