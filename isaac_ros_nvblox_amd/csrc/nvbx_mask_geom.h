// nvbx_mask_geom.h -- the depth-pixel -> mask-pixel projection of ImageMasker::splitImageOnGPU ([U], include/nvblox_hip.h nvbx_split_depth_by_mask),
// shared by the split kernels (convert.hip) and the fused dynamic-mapping front end (dynamics.hip): one arithmetic, bit for bit.
#pragma once
#include "nvbx_mapper.h"

namespace nvbx {

struct Rt { float R[9], t[3]; };
struct MaskGeom { Rt T; float dfu, dfv, dcu, dcv, mfu, mfv, mcu, mcv; int32_t rows, cols, mrows, mcols; };
#ifdef __HIPCC__
// depth pixel (r, c) at depth d -> index of the mask pixel it lands on (or -1) and its depth in the mask camera
__device__ inline int32_t mask_pixel(const MaskGeom& g, int32_t r, int32_t c, float d, float* z_cm) {
  const float rx = (((float)c + 0.5f) - g.dcu) / g.dfu, ry = (((float)r + 0.5f) - g.dcv) / g.dfv;
  float p[3]; apply_rt(g.T.R, g.T.t, d * rx, d * ry, d, p);
  *z_cm = p[2];
  if (p[2] <= 0.0f) return -1;
  const float u = g.mfu * (p[0] / p[2]) + g.mcu, v = g.mfv * (p[1] / p[2]) + g.mcv;
  const int32_t mc = (int32_t)floorf(u), mr = (int32_t)floorf(v);
  if (mc < 0 || mr < 0 || mc >= g.mcols || mr >= g.mrows) return -1;
  return mr * g.mcols + mc;
}
#endif
inline MaskGeom make_mask_geom(const float T_CM_CD[16], const nvbx_camera* dc, const nvbx_camera* mc, int32_t rows, int32_t cols, int32_t mask_rows, int32_t mask_cols) {
  MaskGeom g;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) g.T.R[3 * i + j] = T_CM_CD[4 * i + j]; g.T.t[i] = T_CM_CD[4 * i + 3]; }
  g.dfu = dc->fu; g.dfv = dc->fv; g.dcu = dc->cu; g.dcv = dc->cv; g.mfu = mc->fu; g.mfv = mc->fv; g.mcu = mc->cu; g.mcv = mc->cv;
  g.rows = rows; g.cols = cols; g.mrows = mask_rows; g.mcols = mask_cols;
  return g;
}

}  // namespace nvbx
