/*
 * raymarch_jni.c -- thin JNI shim over the C ABI of libraymarch_hip.so
 * (include/raymarch_hip.h) for the reference's Clojure host.
 *
 * UNVERIFIED IN THIS REPOSITORY'S BUILD IMAGE: the image has no JDK (no jni.h,
 * no javac, no lein), so this file is compiled and exercised nowhere here; it
 * is the binding a maintainer adds on a machine with a JDK:
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
 *       -I../../include raymarch_jni.c -L../../raymarchcl_amd -lraymarch_hip \
 *       -o libraymarch_jni.so
 *
 * Java/Clojure side: class thi.ng.raymarchcl.Native with the static native
 * methods below (see bindings/clojure/thi/ng/raymarchcl/native.clj).  All
 * buffers are DIRECT java.nio buffers, exactly what thi.ng.simplecl hands to
 * JOCL in the reference (core.clj:137-145, io.clj:29-33).  A non-zero return
 * code of the C ABI becomes a RuntimeException carrying rm_last_error().
 */
#include <jni.h>
#include <stdint.h>

#include "raymarch_hip.h"

static jint check(JNIEnv* env, int rc) {
  if (rc != RM_OK) {
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, rm_last_error());
  }
  return rc;
}
static void* addr(JNIEnv* env, jobject buf) { return buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL; }

/* init-renderer's cl/init-state (core.clj:121-128) */
JNIEXPORT jlong JNICALL Java_thi_ng_raymarchcl_Native_create(JNIEnv* env, jclass c, jint device) {
  rm_ctx* ctx = NULL;
  check(env, rm_create(device, &ctx));
  return (jlong)(intptr_t)ctx;
}
JNIEXPORT void JNICALL Java_thi_ng_raymarchcl_Native_destroy(JNIEnv* env, jclass c, jlong h) {
  rm_destroy((rm_ctx*)(intptr_t)h);
}
/* v-buf of vio/load-volume (io.clj:19-33) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_setVolume(JNIEnv* env, jclass c, jlong h,
                                                              jobject vox, jint rx, jint ry, jint rz) {
  return check(env, rm_set_volume((rm_ctx*)(intptr_t)h, (const uint8_t*)addr(env, vox), rx, ry, rz));
}
/* one RenderImage step of the pipeline (core.clj:84-89) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_renderImage(JNIEnv* env, jclass c, jlong h,
                                                                jobject mc, jobject opts, jobject pixels,
                                                                jint n) {
  return check(env, rm_render_image((rm_ctx*)(intptr_t)h, (const float*)addr(env, mc), addr(env, opts),
                                    (float*)addr(env, pixels), n));
}
/* the TonemapImage step (core.clj:91-97) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_tonemapImage(JNIEnv* env, jclass c, jlong h,
                                                                 jobject pixels, jobject opts,
                                                                 jobject argb, jint n) {
  return check(env, rm_tonemap_image((rm_ctx*)(intptr_t)h, (const float*)addr(env, pixels),
                                     addr(env, opts), (uint32_t*)addr(env, argb), n));
}
/* ops/execute-pipeline of make-pipeline (core.clj:76-97, 171) */
JNIEXPORT jint JNICALL Java_thi_ng_raymarchcl_Native_renderFrame(JNIEnv* env, jclass c, jlong h,
                                                                jobject optsArray, jobject mcArray,
                                                                jint iter, jint n, jobject pixelsOut,
                                                                jobject argbOut) {
  return check(env, rm_render_frame((rm_ctx*)(intptr_t)h, addr(env, optsArray),
                                    (const float*)addr(env, mcArray), iter, n,
                                    (float*)addr(env, pixelsOut), (uint32_t*)addr(env, argbOut)));
}
