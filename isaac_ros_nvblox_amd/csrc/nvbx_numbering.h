/* nvbx_numbering.h -- which record of a work list a workgroup takes (plain C: the kernels include it, and tests/test_cabi.py compiles THIS file with gcc
 * and checks the mapping on the CPU). */
#ifndef NVBX_NUMBERING_H_
#define NVBX_NUMBERING_H_
#include <stdint.h>
#if defined(__HIPCC__)
#define NVBX_NUM_HD __host__ __device__ inline
#else
#define NVBX_NUM_HD static inline
#endif
#define NVBX_N_XCD 8        /* == NSH of nvbx_internal.h: workgroups are dispatched round-robin over the 8 XCDs */

// XCD-affine numbering of a work list's records (round 6).  Workgroups go round-robin over the 8 XCDs, each with an L2 of its own, so "workgroup w takes
// record w" hands eight CONSECUTIVE records -- blocks one tile group of the view marking has just appended, neighbours in space whose voxels project onto the
// same patch of the depth / colour image -- to eight different L2s, and every L2 ends up fetching the whole image (PMC: the TSDF update read 7.0 MB for 2.5 MB).
// Here worker w (on XCD w & 7) takes record  q * 64 + (w & 7) * CHUNK + j  with w = q * 64 + j * 8 + (w & 7): runs of CHUNK consecutive records
// share an XCD.  A bijection on [0, n) -- the tail that does not fill a period of 8 * CHUNK keeps its number -- so every record is still taken exactly once, by
// the same code; only WHICH workgroup takes it changes (no result depends on it: one workgroup per block either way).
// Measured (FETCH_SIZE per launch, one box session, tools/chunk_ab.sh; CHUNK 0 / 8 / 16 / 32): k_integrate_tsdf_color 10.3 / 8.6 / 8.0 / 7.7 MB for one camera,
// 77.7 / 62.2 / 58.6 / 56.1 MB for a batch of eight; the launches' durations do not move (8.6-9.2 us, 27.2-27.4 us; 27.9 at 32) -- the chains are what they were,
// the re-reads are gone.  What is left above the image-once figure: consecutive records of the view list come from DIFFERENT tile groups (a group appends only the
// ~3 blocks it is the first to claim), so a run shares an XCD but not always a patch of the image.
#ifndef NVBX_XCD_CHUNK
#define NVBX_XCD_CHUNK 16
#endif
NVBX_NUM_HD int32_t xcd_chunked_c(int32_t w, int32_t n, int32_t C) {       /* C = run length; <= 0: worker w takes record w */
  const int32_t P = NVBX_N_XCD * (C > 0 ? C : 1);
  if (C <= 0 || w >= (n / P) * P) return w;
  const int32_t r = w % P;
  return (w - r) + (r & (NVBX_N_XCD - 1)) * C + (r >> 3);
}
NVBX_NUM_HD int32_t xcd_chunked(int32_t w, int32_t n) { return xcd_chunked_c(w, n, NVBX_XCD_CHUNK); }
#endif
