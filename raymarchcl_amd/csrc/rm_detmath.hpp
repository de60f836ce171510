// rm_detmath.hpp -- deterministic float32 primitives for the gfx950 kernels.
//
// The parity contract of this path is "the same IEEE-754 operation sequence as
// the reference kernel run on an OpenCL CPU device": binary32, round to
// nearest even, no FMA contraction (build with -ffp-contract=off), correctly
// rounded divide and sqrt (hipcc's default -fhip-fp32-correctly-rounded-
// divide-sqrt), denormals preserved.  The three transcendental built-ins the
// reference calls (exp, exp2, pow -- renderer.cl:281,321,322) are evaluated in
// binary64 with +,-,*,/ only and rounded once to float, so that they are the
// same function on every IEEE machine; no ocml / libm calls.
//
// C casts whose result is undefined in the reference source are pinned to the
// x86-64 lowering (what "OpenCL CPU device" means for BASELINE config 1):
//   f2i: truncate; NaN / out of int32 range -> INT_MIN      (cvttss2si r32)
//   f2u: 64-bit truncate, low 32 bits; NaN / |x|>=2^63 -> 0 (cvttss2si r64)
// gfx950's native v_cvt_u32_f32 saturates instead (negatives -> 0) -- which is what
// the reference kernel compiled for a GPU device does: f2u_gpu(), selected per context with
// rm_set_seed_cast(ctx, RM_SEED_CAST_GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define RM_DEV __device__ __forceinline__

namespace rmd {

RM_DEV float fmin_cl(float x, float y) { return y < x ? y : x; }   // OpenCL min(x,y)
RM_DEV float fmax_cl(float x, float y) { return x < y ? y : x; }   // OpenCL max(x,y)
RM_DEV float clamp_cl(float x, float lo, float hi) { return fmin_cl(fmax_cl(x, lo), hi); }
RM_DEV float step_cl(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
RM_DEV float sqrt_rn(float x) { return __builtin_sqrtf(x); }

RM_DEV int32_t f2i(float x) {
  if (!(x >= -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
  return (int32_t)x;
}
RM_DEV uint32_t f2u_gpu(float x) {
  // what v_cvt_u32_f32 does -- truncate, saturate to [0, 2^32 - 1], NaN -> 0 -- written so that no
  // C cast is out of range (the compiler folds the comparisons around one v_cvt_u32_f32)
  // (The instruction itself through inline asm -- asm("v_cvt_u32_f32 %0, %1") -- computes the same value; that
  //  spelling was the round's first reproducer of the compiler fault of DESIGN_HISTORY.md 4c, see
  //  tools/repro_gpucast_fault.sh in the tree of git commit 11be60c.)
  return x > 0.0f ? (x >= 4294967296.0f ? 0xffffffffu : (uint32_t)x) : 0u;
}
RM_DEV uint32_t f2u(float x) {
  if (!(x >= -9223372036854775808.0f && x < 9223372036854775808.0f)) return 0u;
  // |x| < 2^63: split so that only 32-bit hardware conversions are needed.
  // trunc(x) = hi*2^32 + lo exactly (both parts are exact floats); the low 32
  // bits of the two's complement value are lo (mod 2^32) for either sign.
  const float t = __builtin_truncf(x);
  const float a = __builtin_fabsf(t);
  const float hi = __builtin_truncf(a * 2.3283064365386963e-10f);  // floor(a / 2^32), exact
  const float lo = a - hi * 4294967296.0f;                         // exact: a, hi*2^32 share the grid
  const uint32_t m = (uint32_t)lo;                                  // 0 <= lo < 2^32
  return t < 0.0f ? (0u - m) : m;
}
RM_DEV int32_t convert_int_sat(float x) {
  // OpenCL convert_int_sat(float): truncate, saturate, NaN -> 0 -- which is exactly
  // what the hardware's v_cvt_i32_f32 does.  Spelled as the instruction itself so
  // that the compiler cannot treat an out-of-range (int) cast as undefined.
  int32_t r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// x / b, correctly rounded, for several x sharing one divisor: with y = RN(1/b),
//   q = RN(x*y);  r = x - b*q (exact in one fma);  RN(q + r*y) = RN(x/b)
// (Markstein's theorem: holds when the significand of b is not all ones and nothing
// over/underflows -- the range guards; anything else takes the IEEE division.  Checked
// exhaustively over every float x for b = 48, 96 and on 2.6e8 random (x, b) pairs,
// tests/test_det_math.py holds the same check against the built library.)  Three
// instructions per quotient instead of the twelve of a division sequence.
struct Divisor {
  float b, y;
  bool ok;
};
RM_DEV Divisor make_divisor(float b) {
  Divisor d;
  d.b = b;
  d.y = 1.0f / b;
  const float ab = __builtin_fabsf(b);
  d.ok = (ab >= 0x1p-30f) & (ab <= 0x1p30f) & ((__float_as_uint(b) & 0x7fffffu) != 0x7fffffu);
  return d;
}
RM_DEV float div_by(float x, const Divisor& d) {
  const float ax = __builtin_fabsf(x);
  if (d.ok & (ax >= 0x1p-90f) & (ax <= 0x1p90f)) {
    const float q = x * d.y;
    const float r = __builtin_fmaf(-d.b, q, x);
    return __builtin_fmaf(r, d.y, q);
  }
  return x / d.b;
}

RM_DEV double bits2d(uint64_t u) { return __longlong_as_double((long long)u); }
RM_DEV uint64_t d2bits(double d) { return (uint64_t)__double_as_longlong(d); }

// 2^z rounded to float; same algorithm and constants as oracle/cl_scalar.h.
RM_DEV float exp2_core(double z) {
  if (z != z) return (float)z;
  if (z >= 128.0) return __builtin_inff();
  if (z <= -151.0) return 0.0f;
  const double magic = 0x1.8p52;
  const double kd = (z + magic) - magic;
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
  const long long k = (long long)kd;
  const double scale = bits2d((uint64_t)(k + 1023) << 52);
  return (float)(p * scale);
}
RM_DEV float exp2_det(float x) { return exp2_core((double)x); }
RM_DEV float exp_det(float x) { return exp2_core((double)x * 0x1.71547652b82fep+0); }

RM_DEV double log2_pos(double x) {
  const uint64_t b = d2bits(x);
  long long e = (long long)((b >> 52) & 0x7ff) - 1023;
  double m = bits2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
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
RM_DEV float pow_det(float x, float y) {
  if (x != x || y != y) return x + y;
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
  if (x < 0.0f) return __builtin_nanf("");
  if (x == __builtin_inff()) return y > 0.0f ? x : 0.0f;
  return exp2_core((double)y * log2_pos((double)x));
}

}  // namespace rmd
