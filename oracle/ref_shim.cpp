// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE (not product code).
//
// Links the UNMODIFIED reference kernel object (clang -x cl of
// /root/reference/resources/renderer.cl, built by oracle/Makefile into
// oracle/_ref/) into a host shared library:
//   * supplies the 22 OpenCL C built-ins the object leaves undefined, as thin
//     vector wrappers over the scalar semantics in oracle/cl_scalar.h
//     (OpenCL 1.2 spec definitions; see that header for why a shim is needed),
//   * supplies get_global_id() from a thread-local, and
//   * exports a C entry point that runs the reference's RenderImage /
//     TonemapImage (renderer.cl:478-508) for a range of work-item ids, which
//     is what an OpenCL CPU device's NDRange would do.
//
// Must be compiled with clang++ (ext_vector_type mangling == OpenCL floatN)
// and -ffp-contract=off.  No <cmath>: the global names exp/pow/sqrt/... are
// defined here with OpenCL's C++-mangled signatures.
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include "cl_scalar.h"

typedef float f3 __attribute__((ext_vector_type(3)));
typedef int i3 __attribute__((ext_vector_type(3)));

static thread_local size_t tl_gid = 0;
#define SHIM __attribute__((visibility("hidden")))

SHIM size_t get_global_id(unsigned) { return tl_gid; }

SHIM f3 convert_float3(i3 v) { return (f3){(float)v.x, (float)v.y, (float)v.z}; }
SHIM i3 convert_int3_sat(f3 v) {
  return (i3){cl_convert_int_sat(v.x), cl_convert_int_sat(v.y), cl_convert_int_sat(v.z)};
}
SHIM float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
#ifdef RM_SHIM_LIBM
// Second build (oracle/_ref/libref_oracle_libm.so): the three transcendental built-ins come
// from the host's libm instead of cl_scalar.h -- what a CPU OpenCL runtime would most likely
// call.  Used only to MEASURE how far the deterministic definitions are from a libm-backed
// reference (tools/pin_report.py); never the parity oracle.  (Declared by hand: no <cmath>.)
extern "C" float expf(float);
extern "C" float exp2f(float);
extern "C" float powf(float, float);
SHIM float exp(float x) { return expf(x); }
SHIM float exp2(float x) { return exp2f(x); }
SHIM float pow(float x, float y) { return powf(x, y); }
#else
SHIM float exp(float x) { return cl_exp(x); }
SHIM float exp2(float x) { return cl_exp2(x); }
SHIM float pow(float x, float y) { return cl_pow(x, y); }
#endif
SHIM float fabs(float x) { return cl_fabs(x); }
SHIM float sqrt(float x) { return cl_sqrt(x); }
SHIM float mad(float a, float b, float c) { return cl_mad(a, b, c); }
SHIM f3 mad(f3 a, f3 b, f3 c) {
  return (f3){cl_mad(a.x, b.x, c.x), cl_mad(a.y, b.y, c.y), cl_mad(a.z, b.z, c.z)};
}
SHIM float max(float a, float b) { return cl_max(a, b); }
SHIM f3 max(f3 a, f3 b) { return (f3){cl_max(a.x, b.x), cl_max(a.y, b.y), cl_max(a.z, b.z)}; }
SHIM float min(float a, float b) { return cl_min(a, b); }
SHIM f3 min(f3 a, f3 b) { return (f3){cl_min(a.x, b.x), cl_min(a.y, b.y), cl_min(a.z, b.z)}; }
SHIM f3 mix(f3 a, f3 b, f3 t) {
  return (f3){cl_mix(a.x, b.x, t.x), cl_mix(a.y, b.y, t.y), cl_mix(a.z, b.z, t.z)};
}
SHIM f3 mix(f3 a, f3 b, float t) {
  return (f3){cl_mix(a.x, b.x, t), cl_mix(a.y, b.y, t), cl_mix(a.z, b.z, t)};
}
SHIM float step(float e, float x) { return cl_step(e, x); }
SHIM float clamp(float x, float lo, float hi) { return cl_clamp(x, lo, hi); }
SHIM f3 cross(f3 a, f3 b) {
  return (f3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
SHIM float length(f3 v) { return cl_sqrt(dot(v, v)); }
SHIM f3 normalize(f3 v) {
  if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
  const float s = 1.0f / cl_sqrt(dot(v, v));
  return (f3){v.x * s, v.y * s, v.z * s};
}

// The two reference kernels (C linkage in the .cl object).
extern "C" {
SHIM void RenderImage(const unsigned char* voxels, const void* mcSamples, const void* opts,
                      void* pixels, int n);
SHIM void TonemapImage(const void* pixels, void* opts, unsigned* rgba, int n);

// Run work-items id0 <= id < id1 of RenderImage on the calling thread.
__attribute__((visibility("default"))) void ref_render_image(
    const unsigned char* voxels, const float* mc, const void* opts544, float* pixels, int n,
    int id0, int id1) {
  for (int id = id0; id < id1; ++id) {
    tl_gid = (size_t)id;
    RenderImage(voxels, mc, opts544, pixels, n);
  }
}
// The same over `threads` host threads (an OpenCL CPU device spreads an NDRange over its
// cores): chunks of 256 work-items handed out through an atomic counter.  pthreads, not
// <thread>: no C++ math headers may enter this file.
struct MtJob {
  const unsigned char* voxels; const float* mc; const void* opts; float* pixels;
  int n, id1; int next;
};
static void* mt_worker(void* arg) {
  MtJob* j = (MtJob*)arg;
  for (;;) {
    const int lo = __atomic_fetch_add(&j->next, 256, __ATOMIC_RELAXED);
    if (lo >= j->id1) break;
    const int hi = lo + 256 < j->id1 ? lo + 256 : j->id1;
    for (int id = lo; id < hi; ++id) {
      tl_gid = (size_t)id;
      RenderImage(j->voxels, j->mc, j->opts, j->pixels, j->n);
    }
  }
  return 0;
}
__attribute__((visibility("default"))) void ref_render_image_mt(
    const unsigned char* voxels, const float* mc, const void* opts544, float* pixels, int n,
    int id0, int id1, int threads) {
  MtJob job = {voxels, mc, opts544, pixels, n, id1, id0};
  if (threads < 1) threads = 1;
  if (threads > 512) threads = 512;
  pthread_t tid[512];
  int started = 0;
  for (int t = 0; t < threads - 1; t++)
    if (pthread_create(&tid[started], 0, mt_worker, &job) == 0) started++;
  mt_worker(&job);
  for (int t = 0; t < started; t++) pthread_join(tid[t], 0);
}
// Test hook: evaluate built-in `op` -- the very functions the reference object is linked
// against -- on `count` argument tuples.  a, b, c: count x 3 floats for vector arguments,
// count floats for scalar ones (convert_float3 reads a as count x 3 int32); out likewise.
//   0 convert_float3  1 convert_int3_sat (out: int32)  2 dot  3 exp  4 exp2  5 pow  6 fabs  7 sqrt
//   8 mad(f,f,f)  9 mad(f3,f3,f3)  10 max(f,f)  11 max(f3,f3)  12 min(f,f)  13 min(f3,f3)
//   14 mix(f3,f3,f3)  15 mix(f3,f3,f)  16 step  17 clamp  18 cross  19 length  20 normalize
//   21 get_global_id (out[0] = the id set by a = {id})
__attribute__((visibility("default"))) int ref_shim_eval(int op, int count, const float* a, const float* b,
                                                         const float* c, float* out) {
#define V3(p, i) ((f3){(p)[3 * (i)], (p)[3 * (i) + 1], (p)[3 * (i) + 2]})
#define ST3(i, v) do { const f3 v_ = (v); out[3 * (i)] = v_.x; out[3 * (i) + 1] = v_.y; out[3 * (i) + 2] = v_.z; } while (0)
  for (int i = 0; i < count; i++) {
    switch (op) {
      case 0: {
        const int32_t* ia = (const int32_t*)a;
        ST3(i, convert_float3((i3){ia[3 * i], ia[3 * i + 1], ia[3 * i + 2]}));
        break;
      }
      case 1: {
        const i3 r = convert_int3_sat(V3(a, i));
        int32_t* io = (int32_t*)out;
        io[3 * i] = r.x; io[3 * i + 1] = r.y; io[3 * i + 2] = r.z;
        break;
      }
      case 2: out[i] = dot(V3(a, i), V3(b, i)); break;
      case 3: out[i] = exp(a[i]); break;
      case 4: out[i] = exp2(a[i]); break;
      case 5: out[i] = pow(a[i], b[i]); break;
      case 6: out[i] = fabs(a[i]); break;
      case 7: out[i] = sqrt(a[i]); break;
      case 8: out[i] = mad(a[i], b[i], c[i]); break;
      case 9: ST3(i, mad(V3(a, i), V3(b, i), V3(c, i))); break;
      case 10: out[i] = max(a[i], b[i]); break;
      case 11: ST3(i, max(V3(a, i), V3(b, i))); break;
      case 12: out[i] = min(a[i], b[i]); break;
      case 13: ST3(i, min(V3(a, i), V3(b, i))); break;
      case 14: ST3(i, mix(V3(a, i), V3(b, i), V3(c, i))); break;
      case 15: ST3(i, mix(V3(a, i), V3(b, i), c[i])); break;
      case 16: out[i] = step(a[i], b[i]); break;
      case 17: out[i] = clamp(a[i], b[i], c[i]); break;
      case 18: ST3(i, cross(V3(a, i), V3(b, i))); break;
      case 19: out[i] = length(V3(a, i)); break;
      case 20: ST3(i, normalize(V3(a, i))); break;
      case 21: tl_gid = (size_t)a[i]; out[i] = (float)get_global_id(0); break;
      default: return -1;
    }
  }
#undef V3
#undef ST3
  return 0;
}
__attribute__((visibility("default"))) void ref_tonemap_image(const float* pixels,
                                                              const void* opts544,
                                                              uint32_t* argb, int n, int id0,
                                                              int id1) {
  for (int id = id0; id < id1; ++id) {
    tl_gid = (size_t)id;
    TonemapImage(pixels, (void*)opts544, argb, n);
  }
}
}
