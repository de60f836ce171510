// rm_math.hpp -- the two arithmetic contracts of the render path, as policy types.
//
// The reference kernel (renderer.cl) leaves the value of its 21 math built-ins -- and of a few C
// casts that are undefined for out-of-range values -- to the OpenCL device it is built for.  A
// kernel that must reproduce the reference's pixels has to say WHICH device; there are two that
// can be checked here, and the kernels are templates on the choice (Tracer<.., M>):
//
//   MathX86<CAST>  "OpenCL CPU device" (BASELINE config 1): the built-ins as the OpenCL 1.2
//                  specification defines them operation by operation (mad unfused, min/max by
//                  comparison, normalize = v * (1 / sqrt(dot)), exp / exp2 / pow correctly
//                  rounded via binary64 -- rm_detmath.hpp), (int)/(uint) casts as x86-64 lowers
//                  them.  Checked bit for bit against the CPU oracle (oracle/rm_restate.c, itself
//                  bit-identical to the unmodified renderer.cl compiled for x86-64).
//                  CAST = 1: the same arithmetic with the GPU lowering of the (uint) seed casts.
//
//   MathOcl        "this GPU": the built-ins ARE the functions of ROCm's OpenCL built-in library
//                  (/opt/rocm/amdgcn/bitcode/opencl.bc -> ocml / ockl), linked into this code
//                  object by the very symbols the reference kernel links against when ROCm's
//                  OpenCL compiler builds it for gfx950 (mad = fma, min/max = minnum/maxnum,
//                  clamp = med3, normalize through v_rsq_f32, ocml's exp / exp2 / pow, ...);
//                  casts as gfx950 lowers them (v_cvt_i32_f32 / v_cvt_u32_f32: saturating).
//                  Checked bit for bit ON THE GPU against oracle/_ref/renderer_gfx950_strict.hsaco
//                  = the unmodified renderer.cl built with -ffp-contract=off and correctly rounded
//                  divide / sqrt (tests/test_gpu_device_contract.py).
//
//   MathOclT<true> "this GPU, as ROCm's OpenCL compiler builds the reference BY DEFAULT" (MathOclDef): MathOcl plus
//                  the two things clang's OpenCL defaults add to the reference source -- -ffp-contract=on fuses an
//                  a*b+c written INSIDE ONE EXPRESSION into llvm.fmuladd (v_fma_f32 on this chip) at 15 places of
//                  renderer.cl (M::fuse / M::fuse3 / M::nfuse3 below name each), and `/` carries !fpmath 2.5 ulp,
//                  which AMDGPUCodeGenPrepare lowers to frexp / v_rcp_f32 / ldexp (M::div, M::inv).  Checked bit
//                  for bit ON THE GPU against oracle/_ref/renderer_gfx950_default.hsaco = the unmodified
//                  renderer.cl built with NO options; within 1e-4 of the reference's own -cl-fast-relaxed-math
//                  build on 100.0000 % of the pixels of every BASELINE configuration (tests/test_gpu_device_contract.py,
//                  tests/test_gpu_pin_gfx950.py; profiles/r06_pin_gfx950.txt).
//
// Everything that is not a built-in call in the reference source -- +, -, *, / and comparisons
// in source order -- is the same plain float32 code for all of them, EXCEPT at the places the
// source spells a*b+c in one expression or divides: those go through the policy
// (fuse/fuse3/nfuse3/div/inv: two roundings and IEEE division for the first two contracts).
#pragma once
#include "rm_detmath.hpp"

namespace rmk {

struct v3 { float x, y, z; };
RM_DEV v3 V(float x, float y, float z) { return v3{x, y, z}; }
RM_DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
RM_DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
RM_DEV v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
RM_DEV v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
RM_DEV v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
RM_DEV v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
// x, through an identity lane permutation (v_mov_b32_dpp quad_perm:[0,1,2,3]): the same bits, but a value the optimiser
// cannot look through -- what is computed from it stays where it is written (see MathOclT<true>::slab_dir)
RM_DEV float opaque_copy(float x) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(x), 0xE4, 0xf, 0xf, false));
}
// a*s + c written with * and + in the reference source (NOT its mad() built-in): two roundings
// per component under either contract
RM_DEV v3 muladd(v3 a, float s, v3 c) { return V(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }

// ---------------------------------------------------------------------------------------------
template <int CAST>
struct MathX86 {
  static constexpr bool kDevice = false;
  // a*b + c / a*s + c / c - a*s written with * and + INSIDE ONE EXPRESSION of the reference source (not its mad()
  // built-in), and the source's `/`: two roundings, IEEE division under this contract
  RM_DEV static float fuse(float a, float b, float c) { return a * b + c; }
  RM_DEV static v3 fuse3(v3 a, float s, v3 c) { return V(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }
  RM_DEV static v3 nfuse3(v3 a, float s, v3 c) { return V(c.x - a.x * s, c.y - a.y * s, c.z - a.z * s); }
  RM_DEV static float div(float a, float b) { return a / b; }
  RM_DEV static float inv(float x) { return 1.0f / x; }
  RM_DEV static v3 slab_dir(v3 d) { return d; }
  static constexpr bool kExactDiv = true;  // `/` is the IEEE quotient (rmd::div_by may stand in for it)
  RM_DEV static float mad(float a, float b, float c) { return a * b + c; }
  RM_DEV static v3 mads(v3 a, float s, v3 c) { return V(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }
  RM_DEV static v3 madv(v3 a, v3 b, v3 c) { return V(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z); }
  RM_DEV static float mix(float a, float b, float t) { return a + (b - a) * t; }
  RM_DEV static v3 mixs(v3 a, v3 b, float t) { return V(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)); }
  RM_DEV static float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
  RM_DEV static v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
  }
  RM_DEV static v3 normalize(v3 v) {
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
    const float s = 1.0f / rmd::sqrt_rn(dot(v, v));
    return v * s;
  }
  RM_DEV static float length(v3 v) { return rmd::sqrt_rn(dot(v, v)); }
  RM_DEV static float fmin(float x, float y) { return rmd::fmin_cl(x, y); }
  RM_DEV static float fmax(float x, float y) { return rmd::fmax_cl(x, y); }
  RM_DEV static float clamp(float x, float lo, float hi) { return rmd::clamp_cl(x, lo, hi); }
  RM_DEV static float step(float edge, float x) { return rmd::step_cl(edge, x); }
  RM_DEV static float sqrt(float x) { return rmd::sqrt_rn(x); }
  RM_DEV static float exp(float x) { return rmd::exp_det(x); }
  RM_DEV static float exp2(float x) { return rmd::exp2_det(x); }
  RM_DEV static float pow(float x, float y) { return rmd::pow_det(x, y); }
  RM_DEV static int to_int(float x) { return rmd::f2i(x); }  // (int)x
  // (uint)x of a seed expression (renderer.cl:267, 334, 471, 472)
  RM_DEV static uint32_t seed(float x) { return CAST == 0 ? rmd::f2u(x) : rmd::f2u_gpu(x); }
  // one component of convert_int3_sat: truncate, saturate, NaN -> 0 (= v_cvt_i32_f32)
  RM_DEV static int cell(float x) { return rmd::convert_int_sat(x); }
  RM_DEV static void cell3(float tx, float ty, float tz, int& qx, int& qy, int& qz) {
    qx = cell(tx); qy = cell(ty); qz = cell(tz);
  }
  // the accelerated walk converts with cell(): nothing to guard under this contract
  RM_DEV static void walk_guard(v3, v3, int&) {}
};

// ---------------------------------------------------------------------------------------------
// ROCm's OpenCL built-in library, by the symbols the reference kernel leaves undefined (the list
// is `nm` of the reference object).  Declared with ext-vector types and the OpenCL manglings;
// defined by opencl.bc, which _native.py adds to the device libraries of the link.
typedef float cl_f3 __attribute__((ext_vector_type(3)));
typedef int cl_i3 __attribute__((ext_vector_type(3)));
__device__ cl_f3 ocl_normalize(cl_f3) __asm__("_Z9normalizeDv3_f");
__device__ float ocl_length(cl_f3) __asm__("_Z6lengthDv3_f");
__device__ float ocl_dot(cl_f3, cl_f3) __asm__("_Z3dotDv3_fS_");
__device__ cl_f3 ocl_cross(cl_f3, cl_f3) __asm__("_Z5crossDv3_fS_");
__device__ float ocl_mad(float, float, float) __asm__("_Z3madfff");
__device__ cl_f3 ocl_mad3(cl_f3, cl_f3, cl_f3) __asm__("_Z3madDv3_fS_S_");
__device__ cl_f3 ocl_mix3(cl_f3, cl_f3, cl_f3) __asm__("_Z3mixDv3_fS_S_");
__device__ cl_f3 ocl_mix3s(cl_f3, cl_f3, float) __asm__("_Z3mixDv3_fS_f");
__device__ float ocl_min(float, float) __asm__("_Z3minff");
__device__ float ocl_max(float, float) __asm__("_Z3maxff");
__device__ cl_f3 ocl_min3(cl_f3, cl_f3) __asm__("_Z3minDv3_fS_");
__device__ cl_f3 ocl_max3(cl_f3, cl_f3) __asm__("_Z3maxDv3_fS_");
__device__ float ocl_clamp(float, float, float) __asm__("_Z5clampfff");
__device__ float ocl_step(float, float) __asm__("_Z4stepff");
__device__ float ocl_exp(float) __asm__("_Z3expf");
__device__ float ocl_exp2(float) __asm__("_Z4exp2f");
__device__ float ocl_pow(float, float) __asm__("_Z3powff");
__device__ float ocl_sqrt(float) __asm__("_Z4sqrtf");
__device__ float ocl_fabs(float) __asm__("_Z4fabsf");
__device__ cl_i3 ocl_convert_int3_sat(cl_f3) __asm__("_Z16convert_int3_satDv3_f");
__device__ cl_f3 ocl_convert_float3(cl_i3) __asm__("_Z14convert_float3Dv3_i");

// FUSED: false = the reference built with -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt ("strict"),
//        true  = built with no options ("default"): contraction inside expressions + 2.5-ulp division
template <bool FUSED>
struct MathOclT {
  static constexpr bool kDevice = true;
  static constexpr bool kExactDiv = !FUSED;
  // clang -ffp-contract=on: a*b + c in one expression -> llvm.fmuladd -> v_fma_f32 / v_fmac_f32 (one rounding)
  RM_DEV static float fuse(float a, float b, float c) { return FUSED ? __builtin_fmaf(a, b, c) : a * b + c; }
  RM_DEV static v3 fuse3(v3 a, float s, v3 c) { return V(fuse(a.x, s, c.x), fuse(a.y, s, c.y), fuse(a.z, s, c.z)); }
  // c - a*s -> fmuladd(-a, s, c)
  RM_DEV static v3 nfuse3(v3 a, float s, v3 c) {
    if (FUSED) return V(__builtin_fmaf(-a.x, s, c.x), __builtin_fmaf(-a.y, s, c.y), __builtin_fmaf(-a.z, s, c.z));
    return V(c.x - a.x * s, c.y - a.y * s, c.z - a.z * s);
  }
  // OpenCL's `/` without -cl-fp32-correctly-rounded-divide-sqrt: fdiv !fpmath 2.5, which the gfx950 back end
  // (f32 denormals on) expands to ldexp(frexp_mant(a) * rcp(frexp_mant(b)), frexp_exp(a) - frexp_exp(b)),
  // and 1.0f / x to ldexp(rcp(frexp_mant(x)), -frexp_exp(x))  (AMDGPUCodeGenPrepare emitFrexpDiv / emitRcpIEEE1ULP)
  RM_DEV static float div(float a, float b) {
    if (!FUSED) return a / b;
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(b));
    const float q = __builtin_amdgcn_frexp_mantf(a) * r;
    return __builtin_amdgcn_ldexpf(q, __builtin_amdgcn_frexp_expf(a) - __builtin_amdgcn_frexp_expf(b));
  }
  RM_DEV static float inv(float x) {
    if (!FUSED) return 1.0f / x;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(x)), -__builtin_amdgcn_frexp_expf(x));
  }
  // The divisor of the slab test (renderer.cl:154-155) as the test sees it.  The 2.5-ulp division splits into a part
  // that depends on the divisor alone -- rcp(frexp_mant(d)), frexp_exp(d) -- and a ray's direction is the same in
  // every estimate of its march, so the optimiser hoists six values per ray out of the march loop and keeps them in
  // registers across every walk of the ray: 70 spilled VGPRs instead of 39, +5 % frame time at config 2.  (An IEEE
  // division has no such part.)  Three identity moves per slab test keep them inside it.
  RM_DEV static v3 slab_dir(v3 d) { return FUSED ? V(opaque_copy(d.x), opaque_copy(d.y), opaque_copy(d.z)) : d; }
  RM_DEV static cl_f3 v(v3 a) { cl_f3 r = {a.x, a.y, a.z}; return r; }
  RM_DEV static v3 u(cl_f3 a) { return V(a.x, a.y, a.z); }
  RM_DEV static cl_f3 splat(float s) { cl_f3 r = {s, s, s}; return r; }
  RM_DEV static float mad(float a, float b, float c) { return ocl_mad(a, b, c); }
  RM_DEV static v3 mads(v3 a, float s, v3 c) { return u(ocl_mad3(v(a), splat(s), v(c))); }
  RM_DEV static v3 madv(v3 a, v3 b, v3 c) { return u(ocl_mad3(v(a), v(b), v(c))); }
  // (the reference calls mix() on float3 only; a component of it)
  RM_DEV static float mix(float a, float b, float t) { return ocl_mix3s(splat(a), splat(b), t).x; }
  RM_DEV static v3 mixs(v3 a, v3 b, float t) { return u(ocl_mix3s(v(a), v(b), t)); }
  RM_DEV static float dot(v3 a, v3 b) { return ocl_dot(v(a), v(b)); }
  RM_DEV static v3 cross(v3 a, v3 b) { return u(ocl_cross(v(a), v(b))); }
  RM_DEV static v3 normalize(v3 a) { return u(ocl_normalize(v(a))); }
  RM_DEV static float length(v3 a) { return ocl_length(v(a)); }
  RM_DEV static float fmin(float x, float y) { return ocl_min(x, y); }
  RM_DEV static float fmax(float x, float y) { return ocl_max(x, y); }
  RM_DEV static float clamp(float x, float lo, float hi) { return ocl_clamp(x, lo, hi); }
  RM_DEV static float step(float edge, float x) { return ocl_step(edge, x); }
  RM_DEV static float sqrt(float x) { return ocl_sqrt(x); }
  RM_DEV static float exp(float x) { return ocl_exp(x); }
  RM_DEV static float exp2(float x) { return ocl_exp2(x); }
  RM_DEV static float pow(float x, float y) { return ocl_pow(x, y); }
  // (int)x as the reference object does it on this chip: v_cvt_i32_f32 (saturates, NaN -> 0)
  RM_DEV static int to_int(float x) { return rmd::convert_int_sat(x); }
  RM_DEV static uint32_t seed(float x) { return rmd::f2u_gpu(x); }
  // convert_int3_sat of the library: max / min / fptosi / two selects per component -- equal to
  // v_cvt_i32_f32 for every input EXCEPT NaN (library: INT_MIN, instruction: 0)
  RM_DEV static void cell3(float tx, float ty, float tz, int& qx, int& qy, int& qz) {
    cl_f3 t = {tx, ty, tz};
    const cl_i3 q = ocl_convert_int3_sat(t);
    qx = q.x; qy = q.y; qz = q.z;
  }
  // The accelerated walk converts with the bare instruction (one VALU per component instead of
  // seven).  Its samples are p, p + delta, p + 2 delta, ... : finite operands never sum to NaN,
  // so only a NaN already in p or delta can reach a conversion -- where the reference (library
  // conversion -> INT_MIN -> outside the grid) ends the walk: at sample 0 for a NaN in p, at
  // sample 1 for a NaN in delta.  Cut the sample budget accordingly, once per walk.
  RM_DEV static int cell(float x) { return rmd::convert_int_sat(x); }
  RM_DEV static void walk_guard(v3 p, v3 delta, int& steps) {
    const float sp = (p.x + p.y) + p.z, sd = (delta.x + delta.y) + delta.z;  // NaN iff a component is (or inf - inf)
    if (sd != sd) steps = steps < 1 ? steps : 1;
    if (sp != sp) steps = 0;
  }
};
using MathOcl = MathOclT<false>;
using MathOclDef = MathOclT<true>;

// The contract of a context as the kernels are instantiated on it (ARITH): one table, used by every launcher
//   0  MathX86<0>  OpenCL CPU device                  1  MathX86<1>  the same, GPU lowering of the seed casts
//   2  MathOcl     this GPU, strict reference build   3  MathOclDef  this GPU, default reference build
constexpr int kArithCount = 4;
template <int ARITH> struct ArithOf;
template <> struct ArithOf<0> { using type = MathX86<0>; };
template <> struct ArithOf<1> { using type = MathX86<1>; };
template <> struct ArithOf<2> { using type = MathOcl; };
template <> struct ArithOf<3> { using type = MathOclDef; };

}  // namespace rmk
