/*
 * jni.h -- TEST-ONLY declaration guard (tests/test_jni_shim.py).
 *
 * NOT the JDK's header: the build image has no JDK.  It declares, under the names and
 * signatures of the Java Native Interface specification (chapter 4, "JNI Functions"), exactly
 * the types and the six interface functions bindings/jni/raymarch_jni.c uses, so that the
 * shim can be compiled and its Java_* entry points driven from C (test/harness.c supplies the
 * function table).  The real JNINativeInterface_ has ~230 slots in a fixed order; this struct
 * holds only the used ones, so objects compiled against it are NOT loadable by a JVM -- a
 * maintainer rebuilds against $JAVA_HOME/include/jni.h (same source, no change).
 */
#ifndef RM_TEST_JNI_H
#define RM_TEST_JNI_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef struct rm_test_jobject_* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef uint8_t jboolean;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv* env, const char* name);
  jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
  const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* isCopy);
  void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
};

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#endif
