/*
 * raymarch_jni.c -- thin JNI shim over the C ABI of libraymarch_hip.so
 * (include/raymarch_hip.h) for the reference's Clojure host.
 *
 * The build image has no JDK (no jni.h, no javac, no lein).  What IS checked here:
 * tests/test_jni_shim.py compiles this file against bindings/jni/test/jni.h -- a hand-written
 * declaration guard holding the handful of JNI declarations the shim uses -- and drives every
 * Java_* entry point from a C harness through a stand-in JNIEnv function table
 * (bindings/jni/test/harness.c), on the GPU, comparing with direct C-ABI calls.  On a machine
 * with a JDK a maintainer builds it against the real header:
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
 *       -I../../include raymarch_jni.c -L../../raymarchcl_amd -lraymarch_hip \
 *       -o libraymarch_jni.so
 *
 * Java side: class thi.ng.raymarchcl.Native (bindings/java/thi/ng/raymarchcl/Native.java) with
 * the static native methods below; Clojure side: bindings/clojure/thi/ng/raymarchcl/native.clj.
 * All buffers are DIRECT java.nio buffers, exactly what thi.ng.simplecl hands to JOCL in the
 * reference (core.clj:137-145, io.clj:29-33).  A non-zero return code of the C ABI becomes a
 * RuntimeException carrying rm_last_error().
 */
#include <jni.h>
#include <stdint.h>
#include <stddef.h>

#include "raymarch_hip.h"

static jint check(JNIEnv* env, int rc) {
  if (rc != RM_OK) {
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, rm_last_error());
  }
  return rc;
}
static void* throw_iae(JNIEnv* env, const char* what) {
  jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
  if (ex) (*env)->ThrowNew(env, ex, what);
  return NULL;
}
/* A REQUIRED direct ByteBuffer that must hold at least `bytes` (every buffer of Native.java is a
 * ByteBuffer: GetDirectBufferCapacity counts elements of the buffer's own type).  Null, not direct
 * or too small: IllegalArgumentException, NULL returned, no native call is made. */
static void* addr_of(JNIEnv* env, jobject buf, jlong bytes, const char* what) {
  if (!buf) return throw_iae(env, what);
  void* p = (*env)->GetDirectBufferAddress(env, buf);
  const jlong cap = (*env)->GetDirectBufferCapacity(env, buf);
  if (!p || cap < bytes) return throw_iae(env, what);
  return p;
}
/* an OPTIONAL output buffer: null is allowed (*absent = 1), anything else is checked as above */
static void* opt_addr_of(JNIEnv* env, jobject buf, jlong bytes, const char* what, int* bad) {
  if (!buf) return NULL;
  void* p = addr_of(env, buf, bytes, what);
  if (!p) *bad = 1;
  return p;
}
#define CTX(h) ((rm_ctx*)(intptr_t)(h))

/* init-renderer's cl/select-platform .. cl/init-state (core.clj:121-128) */
JNIEXPORT jlong JNICALL Java_thi_ng_raymarchcl_Native_create(JNIEnv* env, jclass c, jint device) {
  rm_ctx* ctx = NULL;
  (void)c;
  check(env, rm_create(device, &ctx));
  return (jlong)(intptr_t)ctx;
}
/* the same over the first `nDevices` GPUs of the node: frames are tiled over them */
JNIEXPORT jlong JNICALL Java_thi_ng_raymarchcl_Native_createMulti(JNIEnv* env, jclass c, jobject deviceIds,
                                                                  jint nDevices) {
  rm_ctx* ctx = NULL;
  (void)c;
  const int* ids = (const int*)addr_of(env, deviceIds, (jlong)nDevices * 4, "deviceIds: needs a direct ByteBuffer of nDevices ints");
  if (!ids) return 0;
  check(env, rm_create_multi(ids, nDevices, &ctx));
  return (jlong)(intptr_t)ctx;
}
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_deviceCount(JNIEnv* env, jclass c) {
  (void)env; (void)c;
  return rm_device_count();
}
JNIEXPORT void JNICALL Java_thi_ng_raymarchcl_Native_destroy(JNIEnv* env, jclass c, jlong h) {
  (void)env; (void)c;
  rm_destroy(CTX(h));
}
/* v-buf of vio/load-volume (io.clj:19-33) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_setVolume(JNIEnv* env, jclass c, jlong h,
                                                              jobject vox, jint rx, jint ry, jint rz) {
  (void)c;
  const uint8_t* p = (const uint8_t*)addr_of(env, vox, (jlong)rx * ry * rz, "voxels: direct ByteBuffer too small");
  if (!p) return RM_EINVAL;
  return check(env, rm_set_volume(CTX(h), p, rx, ry, rz));
}
/* the volume of the NEXT animation frame (meshvoxel.clj:85-89 make-heatmap-anim -> core.clj:181-213): its tables are built
 * on the library's own stream while the host still works on the current frame; commitStagedVolume makes it resident */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_stageVolume(JNIEnv* env, jclass c, jlong h, jobject vox, jint rx,
                                                                jint ry, jint rz, jint isoVal) {
  (void)c;
  const uint8_t* p = (const uint8_t*)addr_of(env, vox, (jlong)rx * ry * rz, "voxels: direct ByteBuffer too small");
  if (!p) return RM_EINVAL;
  return check(env, rm_stage_volume(CTX(h), p, rx, ry, rz, isoVal));
}
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_commitStagedVolume(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  return check(env, rm_commit_staged_volume(CTX(h)));
}
/* gen/make-gyroid-volume (generators.clj:27-42) on the device; voxelsOut may be null */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_makeGyroidVolume(JNIEnv* env, jclass c, jlong h, jint rx,
                                                                     jint ry, jint rz, jobject voxelsOut) {
  (void)c;
  int bad = 0;
  uint8_t* p = (uint8_t*)opt_addr_of(env, voxelsOut, (jlong)rx * ry * rz, "voxelsOut: direct ByteBuffer too small", &bad);
  if (bad) return RM_EINVAL;
  return check(env, rm_make_gyroid_volume(CTX(h), rx, ry, rz, p));
}
/* one RenderImage step of the pipeline (core.clj:84-89) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_renderImage(JNIEnv* env, jclass c, jlong h,
                                                                jobject mc, jobject opts, jobject pixels,
                                                                jint n) {
  (void)c;
  const float* pm = (const float*)addr_of(env, mc, (jlong)RM_TABLE_FLOATS * 4, "mc: needs 0x4000 float4");
  const void* po = addr_of(env, opts, RM_OPTS_BYTES, "opts: needs 544 bytes");
  float* pp = (float*)addr_of(env, pixels, (jlong)n * 16, "pixels: needs n float4");
  if (!pm || !po || !pp) return RM_EINVAL;
  return check(env, rm_render_image(CTX(h), pm, po, pp, n));
}
/* the TonemapImage step (core.clj:91-97) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_tonemapImage(JNIEnv* env, jclass c, jlong h,
                                                                 jobject pixels, jobject opts,
                                                                 jobject argb, jint n) {
  (void)c;
  const float* pp = (const float*)addr_of(env, pixels, (jlong)n * 16, "pixels: needs n float4");
  const void* po = addr_of(env, opts, RM_OPTS_BYTES, "opts: needs 544 bytes");
  uint32_t* pa = (uint32_t*)addr_of(env, argb, (jlong)n * 4, "argb: needs n ints");
  if (!pp || !po || !pa) return RM_EINVAL;
  return check(env, rm_tonemap_image(CTX(h), pp, po, pa, n));
}
/* ops/execute-pipeline of make-pipeline (core.clj:76-97, 171); pixelsOut / argbOut may be null */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_renderFrame(JNIEnv* env, jclass c, jlong h,
                                                                jobject optsArray, jobject mcArray,
                                                                jint iter, jint n, jobject pixelsOut,
                                                                jobject argbOut) {
  (void)c;
  const void* po = addr_of(env, optsArray, (jlong)iter * RM_OPTS_BYTES, "optsArray: needs iter x 544 bytes");
  const float* pm = (const float*)addr_of(env, mcArray, (jlong)iter * RM_TABLE_FLOATS * 4, "mcArray: needs iter tables");
  int bad = 0;
  float* pp = (float*)opt_addr_of(env, pixelsOut, (jlong)n * 16, "pixelsOut: needs n float4", &bad);
  uint32_t* pa = (uint32_t*)opt_addr_of(env, argbOut, (jlong)n * 4, "argbOut: needs n ints", &bad);
  if (!po || !pm || bad) return RM_EINVAL;
  return check(env, rm_render_frame(CTX(h), po, pm, iter, n, pp, pa));
}
/* device time of the render kernel(s) of the last frame, in milliseconds (< 0: none yet) */
JNIEXPORT jfloat JNICALL Java_thi_ng_raymarchcl_Native_lastFrameMillis(JNIEnv* env, jclass c, jlong h) {
  float ms = -1.0f;
  (void)env; (void)c;
  if (rm_last_frame_timing(CTX(h), &ms, NULL) != RM_OK) return -1.0f;
  return ms;
}
/* gen/generate-scatter-offsets (generators.clj:8-16) with a seed; out: 0x4000 float4 */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_makeScatterTable(JNIEnv* env, jclass c, jlong seed,
                                                                     jobject out) {
  (void)c;
  float* p = (float*)addr_of(env, out, (jlong)RM_TABLE_FLOATS * 4, "out: needs 0x4000 float4");
  if (!p) return RM_EINVAL;
  return check(env, rm_make_scatter_table((uint64_t)seed, p));
}

/* whose results the kernels reproduce (rm_set_contract): 2 = the reference kernel as ROCm's OpenCL compiler builds it for
 * this GPU with no options (the library default), 1 = built with -ffp-contract=off and correctly rounded divide/sqrt,
 * 0 = an OpenCL CPU device */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_setContract(JNIEnv* env, jclass c, jlong h, jint contract) {
  (void)c;
  return check(env, rm_set_contract(CTX(h), contract));
}
/* vio/load-volume (io.clj:19-33) without OpenCL: header of a .vox file -> out = {rx, ry, rz} (3 ints) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_voxInfo(JNIEnv* env, jclass c, jstring path, jobject out3) {
  (void)c;
  int* o = (int*)addr_of(env, out3, 12, "out: needs 3 ints");
  if (!o) return RM_EINVAL;
  if (!path) { throw_iae(env, "path is null"); return RM_EINVAL; }
  const char* p = (*env)->GetStringUTFChars(env, path, NULL);
  if (!p) return RM_EINVAL;
  const int rc = rm_vox_info(p, &o[0], &o[1], &o[2]);
  (*env)->ReleaseStringUTFChars(env, path, p);
  return check(env, rc);
}
/* ... and its bytes into a direct buffer of at least rx*ry*rz bytes */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_voxLoad(JNIEnv* env, jclass c, jstring path, jobject voxelsOut,
                                                            jlong capacity) {
  (void)c;
  uint8_t* o = (uint8_t*)addr_of(env, voxelsOut, capacity, "voxelsOut: direct ByteBuffer smaller than `capacity`");
  if (!o) return RM_EINVAL;
  if (!path) { throw_iae(env, "path is null"); return RM_EINVAL; }
  const char* p = (*env)->GetStringUTFChars(env, path, NULL);
  if (!p) return RM_EINVAL;
  const int rc = rm_vox_load(p, o, (size_t)capacity);
  (*env)->ReleaseStringUTFChars(env, path, p);
  return check(env, rc);
}

/* Page-lock a long-lived direct buffer (capacity bytes) that renderFrame will be given frame after
 * frame (rm_pin_host_buffer); it must stay reachable until unpin / destroy. */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_pin(JNIEnv* env, jclass c, jlong h, jobject buf, jlong bytes) {
  (void)c;
  void* p = addr_of(env, buf, bytes, "buf: direct ByteBuffer smaller than `bytes`");
  if (!p) return RM_EINVAL;
  return check(env, rm_pin_host_buffer(CTX(h), p, (size_t)bytes));
}
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_unpin(JNIEnv* env, jclass c, jlong h, jobject buf) {
  (void)c;
  void* p = addr_of(env, buf, 0, "buf: not a direct ByteBuffer");
  if (!p) return RM_EINVAL;
  return check(env, rm_unpin_host_buffer(CTX(h), p));
}
