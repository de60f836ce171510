/*
 * oracle/cl_scalar.h -- TEST INFRASTRUCTURE (not product code).
 *
 * Scalar float32 semantics of the 22 OpenCL C built-ins that the reference
 * kernel (/root/reference/resources/renderer.cl) calls.  The reference gets
 * them from its OpenCL runtime (un-vendored: thi.ng/simplecl 0.2.2 -> JOCL ->
 * vendor ICD, project.clj:11); no x86 OpenCL built-in library exists in this
 * image, so the definitions below restate the OpenCL 1.2 specification
 * (section 6.12.2 math, 6.12.4 common, 6.12.5 geometric, 6.2.3 conversions):
 *
 *   min(x,y)   = y < x ? y : x            max(x,y)  = x < y ? y : x
 *   clamp      = min(max(x,lo),hi)        step(e,x) = x < e ? 0 : 1
 *   mix(x,y,a) = x + (y-x)*a              mad(a,b,c)= a*b + c  (NOT fused)
 *   dot        = x*x'+y*y'+z*z' summed left to right
 *   length     = sqrt(dot(v,v))           normalize = 0 -> 0, else v*(1/sqrt(dot))
 *   convert_int_sat(float): truncate toward zero, saturate, NaN -> 0
 *   exp/exp2/pow: evaluated in IEEE double with +,-,*,/ only (no libm), then
 *       rounded once to float -- deterministic on every IEEE machine, which is
 *       what lets the x86 oracle and the gfx950 kernel agree bit for bit.
 *       Error < 1e-13 relative before the final rounding, i.e. the result is
 *       the correctly rounded float except for ~1e-6 of arguments.
 *
 * C casts that are undefined/implementation-defined in the reference source
 * ((int)float and (uint)float, renderer.cl:246,267,334,471,472,504-506) are
 * pinned to what clang emits for x86-64 (cvttss2si):
 *   cl_f2i : truncate; NaN / out of range -> INT_MIN
 *   cl_f2u : 64-bit truncate then keep the low 32 bits (wraps for negatives);
 *            NaN / |x| >= 2^63 -> 0
 *
 * Everything here must be compiled with -ffp-contract=off.
 */
#ifndef RM_ORACLE_CL_SCALAR_H
#define RM_ORACLE_CL_SCALAR_H
#include <stdint.h>
#include <string.h>

static inline uint64_t cl_d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double   cl_bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static inline float cl_min(float x, float y) { return y < x ? y : x; }
static inline float cl_max(float x, float y) { return x < y ? y : x; }
static inline float cl_clamp(float x, float lo, float hi) { return cl_min(cl_max(x, lo), hi); }
static inline float cl_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float cl_mix(float x, float y, float a) { return x + (y - x) * a; }
static inline float cl_mad(float a, float b, float c) { return a * b + c; }
static inline float cl_fabs(float x) { return __builtin_fabsf(x); }
/* x86 sqrtss is correctly rounded */
static inline float cl_sqrt(float x) { return __builtin_sqrtf(x); }

static inline int32_t cl_f2i(float x) {
  if (!(x >= -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
  return (int32_t)x;
}
static inline uint32_t cl_f2u(float x) {
  if (!(x >= -9223372036854775808.0f && x < 9223372036854775808.0f)) return 0u;
  return (uint32_t)(uint64_t)(int64_t)x;
}
static inline int32_t cl_convert_int_sat(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}

/* 2^z for a double z, rounded to float.  z = k + r, |r| <= 1/2,
 * 2^r = exp(r ln2) by a degree-14 Taylor polynomial (Horner). */
static inline float cl_exp2_core(double z) {
  if (z != z) return (float)z;
  if (z >= 128.0) return __builtin_inff();
  if (z <= -151.0) return 0.0f;
  const double magic = 0x1.8p52;
  const double kd = (z + magic) - magic; /* round to nearest integer */
  const double t = (z - kd) * 0x1.62e42fefa39efp-1;
  double p = 0x1.93974a8c07c9dp-37;
  p = p * t + 0x1.6124613a86d09p-33;
  p = p * t + 0x1.1eed8eff8d898p-29;
  p = p * t + 0x1.ae64567f544e4p-26;
  p = p * t + 0x1.27e4fb7789f5cp-22;
  p = p * t + 0x1.71de3a556c734p-19;
  p = p * t + 0x1.a01a01a01a01ap-16;
  p = p * t + 0x1.a01a01a01a01ap-13;
  p = p * t + 0x1.6c16c16c16c17p-10;
  p = p * t + 0x1.1111111111111p-7;
  p = p * t + 0x1.5555555555555p-5;
  p = p * t + 0x1.5555555555555p-3;
  p = p * t + 0.5;
  p = p * t + 1.0;
  p = p * t + 1.0;
  const int64_t k = (int64_t)kd;
  const double scale = cl_bits2d((uint64_t)(k + 1023) << 52); /* k in [-151,128] */
  return (float)(p * scale);
}
static inline float cl_exp2(float x) { return cl_exp2_core((double)x); }
static inline float cl_exp(float x) { return cl_exp2_core((double)x * 0x1.71547652b82fep+0); }

/* log2 of a positive finite double: x = 2^e * m, m in (sqrt(1/2), sqrt 2],
 * ln m = 2 atanh((m-1)/(m+1)) by its odd series to f^25. */
static inline double cl_log2_pos(double x) {
  const uint64_t b = cl_d2bits(x);
  int64_t e = (int64_t)((b >> 52) & 0x7ff) - 1023;
  double m = cl_bits2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e = e + 1; }
  const double f = (m - 1.0) / (m + 1.0);
  const double g = f * f;
  double s = 0x1.47ae147ae147bp-4;
  s = s * g + 0x1.642c8590b2164p-4;
  s = s * g + 0x1.8618618618618p-4;
  s = s * g + 0x1.af286bca1af28p-4;
  s = s * g + 0x1.e1e1e1e1e1e1ep-4;
  s = s * g + 0x1.1111111111111p-3;
  s = s * g + 0x1.3b13b13b13b14p-3;
  s = s * g + 0x1.745d1745d1746p-3;
  s = s * g + 0x1.c71c71c71c71cp-3;
  s = s * g + 0x1.2492492492492p-2;
  s = s * g + 0x1.999999999999ap-2;
  s = s * g + 0x1.5555555555555p-1;
  s = s * g + 2.0;
  return (double)e + (s * f) * 0x1.71547652b82fep+0;
}
/* pow(x,y): the reference only calls it with x > 0 and finite y > 0
 * (renderer.cl:320-322); the remaining cases follow C99 F.9.4.4 for that
 * quadrant and return NaN for negative x (no integer-y special casing). */
static inline float cl_pow(float x, float y) {
  if (x != x || y != y) return x + y;
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
  if (x < 0.0f) return __builtin_nanf("");
  if (x == __builtin_inff()) return y > 0.0f ? x : 0.0f;
  return cl_exp2_core((double)y * cl_log2_pos((double)x));
}
#endif
