/* nvbx_arith.h -- the three IEEE operations the kernels spend most VALU cycles on, in the form the compiler would emit minus the
 * parts the operand ranges of this library never need.  Plain C on the host (the CPU oracle includes the sensor model that uses
 * these), device builtins under hipcc.  RESULTS ARE THE CORRECTLY ROUNDED IEEE RESULTS on both sides, so host and device agree bit
 * for bit:
 *   NVBX_DIV(a, b)   device: v_rcp_f32 + the compiler's own Newton / residual FMA chain + v_div_fixup_f32, WITHOUT the two
 *                    v_div_scale_f32 / v_div_fmas_f32 range-scaling steps: exact while b, a / b and the residual a - b q stay in the
 *                    normal range, i.e. for |b| and non-zero |a| within [2^-60, 2^60] (metres, weights, pixel coordinates: always);
 *                    signed zeros, infinities and NaNs go through v_div_fixup_f32 as in the full sequence.  9 instructions for 11.
 *   NVBX_SQRT(x)     device: v_sqrt_f32 + the compiler's two one-ulp residual corrections, WITHOUT the 2^32 pre-scaling of arguments
 *                    below 2^-96 and the class test: exact for x = 0 and x in [2^-96, 2^127).  9 instructions for 15.
 *   NVBX_FMA(a,b,c)  fmaf: one rounding on both sides (-ffp-contract=off stays: nothing is contracted behind the code's back).
 * tests/test_gpu_arith.py checks NVBX_DIV / NVBX_SQRT against numpy's float32 division / square root on 2^24 operand pairs.
 */
#ifndef NVBX_ARITH_H_
#define NVBX_ARITH_H_
#include <math.h>

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline float nvbx_dev_div(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = a * r;
  float rem = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(rem, r, q);
  rem = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(rem, r, q);
  return __builtin_amdgcn_div_fixupf(q, b, a);
}
__device__ inline float nvbx_dev_sqrt(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
  const float s_up = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x);
  const float r_up = __builtin_fmaf(-s_up, s, x);
  float o = (r_dn <= 0.0f) ? s_dn : s;
  o = (r_up > 0.0f) ? s_up : o;
  return o;
}
#define NVBX_DIV(a, b) nvbx_dev_div((a), (b))
#define NVBX_SQRT(x) nvbx_dev_sqrt(x)
/* "does any lane of the wavefront need the rare path": a wave-uniform (scalar) branch the compiler cannot turn into compute-both-
 * and-select, which it otherwise does for short rare paths -- at the price of running them for every voxel */
#define NVBX_ANY_LANE(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)
#define NVBX_KEEP_HERE(x) asm volatile("" : "+v"(x))       /* on the rare path's input: pins the path behind its branch */
#else
#define NVBX_KEEP_HERE(x) ((void)0)
#define NVBX_DIV(a, b) ((a) / (b))
#define NVBX_SQRT(x) sqrtf(x)
#define NVBX_ANY_LANE(cond) (cond)
#endif
#define NVBX_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#endif
