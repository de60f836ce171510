/*
 * harness.c -- drives every Java_thi_ng_raymarchcl_Native_* entry point of the JNI shim from C,
 * through a stand-in JNIEnv function table (tests/test_jni_shim.py builds and runs it on the GPU).
 *
 *   harness <scene.bin> <out.bin>
 *
 * scene.bin: int32 rx, ry, rz, iter, n, then rx*ry*rz voxel bytes, iter*544 option bytes,
 * iter*0x4000*4 floats.  out.bin: n float4 + n ARGB words of Native.renderFrame, then the same
 * from the single-pass entry points (renderImage per pass + tonemapImage), then one int32 per
 * check (1 = passed: expected Java exceptions raised, helpers equal to the C ABI, .vox round trip).
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raymarch_hip.h"

/* what a direct java.nio buffer is to the shim: an address and a capacity */
struct rm_test_jobject_ { void* addr; jlong cap; const char* klass; };

static char g_thrown_class[128];
static char g_thrown_msg[512];
static int g_throws = 0;

static jclass t_FindClass(JNIEnv* env, const char* name) {
  static struct rm_test_jobject_ k;
  (void)env;
  k.klass = name;
  return &k;
}
static jint t_ThrowNew(JNIEnv* env, jclass clazz, const char* msg) {
  (void)env;
  snprintf(g_thrown_class, sizeof g_thrown_class, "%s", clazz && clazz->klass ? clazz->klass : "?");
  snprintf(g_thrown_msg, sizeof g_thrown_msg, "%s", msg ? msg : "");
  g_throws++;
  return 0;
}
static void* t_Addr(JNIEnv* env, jobject b) { (void)env; return b->addr; }
static jlong t_Cap(JNIEnv* env, jobject b) { (void)env; return b->cap; }
/* a java.lang.String to the shim: the stand-in object carries the UTF-8 bytes in `addr` */
static int g_strings_out = 0;
static const char* t_GetUTF(JNIEnv* env, jstring s, jboolean* is_copy) {
  (void)env;
  if (is_copy) *is_copy = 0;
  g_strings_out++;
  return (const char*)s->addr;
}
static void t_ReleaseUTF(JNIEnv* env, jstring s, const char* chars) { (void)env; (void)s; (void)chars; g_strings_out--; }

/* the shim's exports */
jlong Java_thi_ng_raymarchcl_Native_create(JNIEnv*, jclass, jint);
jlong Java_thi_ng_raymarchcl_Native_createMulti(JNIEnv*, jclass, jobject, jint);
jint Java_thi_ng_raymarchcl_Native_deviceCount(JNIEnv*, jclass);
void Java_thi_ng_raymarchcl_Native_destroy(JNIEnv*, jclass, jlong);
jint Java_thi_ng_raymarchcl_Native_setVolume(JNIEnv*, jclass, jlong, jobject, jint, jint, jint);
jint Java_thi_ng_raymarchcl_Native_makeGyroidVolume(JNIEnv*, jclass, jlong, jint, jint, jint, jobject);
jint Java_thi_ng_raymarchcl_Native_stageVolume(JNIEnv*, jclass, jlong, jobject, jint, jint, jint, jint);
jint Java_thi_ng_raymarchcl_Native_commitStagedVolume(JNIEnv*, jclass, jlong);
jint Java_thi_ng_raymarchcl_Native_renderImage(JNIEnv*, jclass, jlong, jobject, jobject, jobject, jint);
jint Java_thi_ng_raymarchcl_Native_tonemapImage(JNIEnv*, jclass, jlong, jobject, jobject, jobject, jint);
jint Java_thi_ng_raymarchcl_Native_renderFrame(JNIEnv*, jclass, jlong, jobject, jobject, jint, jint, jobject, jobject);
jfloat Java_thi_ng_raymarchcl_Native_lastFrameMillis(JNIEnv*, jclass, jlong);
jint Java_thi_ng_raymarchcl_Native_makeScatterTable(JNIEnv*, jclass, jlong, jobject);
jint Java_thi_ng_raymarchcl_Native_setContract(JNIEnv*, jclass, jlong, jint);
jint Java_thi_ng_raymarchcl_Native_voxInfo(JNIEnv*, jclass, jstring, jobject);
jint Java_thi_ng_raymarchcl_Native_voxLoad(JNIEnv*, jclass, jstring, jobject, jlong);
jint Java_thi_ng_raymarchcl_Native_pin(JNIEnv*, jclass, jlong, jobject, jlong);
jint Java_thi_ng_raymarchcl_Native_unpin(JNIEnv*, jclass, jlong, jobject);

static struct rm_test_jobject_ buf(void* p, size_t bytes) {
  struct rm_test_jobject_ b = {p, (jlong)bytes, NULL};
  return b;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const struct JNINativeInterface_ table = {t_FindClass, t_ThrowNew, t_Addr, t_Cap, t_GetUTF, t_ReleaseUTF};
  JNIEnv envp = &table;
  JNIEnv* env = &envp;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[5];
  if (fread(hdr, 4, 5, f) != 5) return 3;
  const int rx = hdr[0], ry = hdr[1], rz = hdr[2], iter = hdr[3], n = hdr[4];
  const size_t nvox = (size_t)rx * ry * rz, nopt = (size_t)iter * RM_OPTS_BYTES, nmc = (size_t)iter * RM_TABLE_FLOATS;
  uint8_t* vox = malloc(nvox);
  uint8_t* opts = malloc(nopt);
  float* mc = malloc(nmc * 4);
  if (fread(vox, 1, nvox, f) != nvox || fread(opts, 1, nopt, f) != nopt || fread(mc, 4, nmc, f) != nmc) return 3;
  fclose(f);
  float* px = calloc((size_t)n * 4, 4);
  uint32_t* argb = calloc(n, 4);
  float* px1 = calloc((size_t)n * 4, 4);
  uint32_t* argb1 = calloc(n, 4);
  int32_t checks[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  if (Java_thi_ng_raymarchcl_Native_deviceCount(env, NULL) < 1) return 4;
  const jlong h = Java_thi_ng_raymarchcl_Native_create(env, NULL, 0);
  if (!h || g_throws) { fprintf(stderr, "create: %s\n", g_thrown_msg); return 4; }
  struct rm_test_jobject_ bvox = buf(vox, nvox), bopts = buf(opts, nopt), bmc = buf(mc, nmc * 4),
                          bpx = buf(px, (size_t)n * 16), bargb = buf(argb, (size_t)n * 4);
  /* error path 0: frame before a volume -> RuntimeException with the library's message */
  g_throws = 0;
  Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, &bopts, &bmc, iter, n, &bpx, &bargb);
  checks[0] = g_throws == 1 && strcmp(g_thrown_class, "java/lang/RuntimeException") == 0 &&
              strstr(g_thrown_msg, "rm_set_volume") != NULL;
  /* error path 1: a direct buffer that is too small -> IllegalArgumentException, no native call */
  g_throws = 0;
  struct rm_test_jobject_ small = buf(vox, nvox - 1);
  Java_thi_ng_raymarchcl_Native_setVolume(env, NULL, h, &small, rx, ry, rz);
  checks[1] = g_throws == 1 && strcmp(g_thrown_class, "java/lang/IllegalArgumentException") == 0;
  g_throws = 0;
  if (Java_thi_ng_raymarchcl_Native_setVolume(env, NULL, h, &bvox, rx, ry, rz) != 0 || g_throws) return 5;
  if (Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, &bopts, &bmc, iter, n, &bpx, &bargb) != 0 || g_throws) {
    fprintf(stderr, "renderFrame: %s\n", g_thrown_msg);
    return 6;
  }
  checks[2] = Java_thi_ng_raymarchcl_Native_lastFrameMillis(env, NULL, h) > 0.0f;
  /* the single-pass entry points: the reference's pipeline step by step (core.clj:81-97) */
  for (int i = 0; i < iter; i++) {
    struct rm_test_jobject_ bo = buf(opts + (size_t)i * RM_OPTS_BYTES, RM_OPTS_BYTES),
                            bm = buf(mc + (size_t)i * RM_TABLE_FLOATS, (size_t)RM_TABLE_FLOATS * 4),
                            bp = buf(px1, (size_t)n * 16);
    if (Java_thi_ng_raymarchcl_Native_renderImage(env, NULL, h, &bm, &bo, &bp, n) != 0) return 7;
  }
  {
    struct rm_test_jobject_ bo = buf(opts, RM_OPTS_BYTES), bp = buf(px1, (size_t)n * 16), ba = buf(argb1, (size_t)n * 4);
    if (Java_thi_ng_raymarchcl_Native_tonemapImage(env, NULL, h, &bp, &bo, &ba, n) != 0) return 8;
  }
  /* two ranks inside the library through createMulti (both on device 0) == the same frame */
  {
    int32_t ids[2] = {0, 0};
    struct rm_test_jobject_ bids = buf(ids, 8);
    const jlong hm = Java_thi_ng_raymarchcl_Native_createMulti(env, NULL, &bids, 2);
    float* pxm = calloc((size_t)n * 4, 4);
    struct rm_test_jobject_ bpm = buf(pxm, (size_t)n * 16);
    if (hm && Java_thi_ng_raymarchcl_Native_setVolume(env, NULL, hm, &bvox, rx, ry, rz) == 0 &&
        Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, hm, &bopts, &bmc, iter, n, &bpm, NULL) == 0)
      checks[3] = memcmp(pxm, px, (size_t)n * 16) == 0;
    Java_thi_ng_raymarchcl_Native_destroy(env, NULL, hm);
    free(pxm);
  }
  /* scatter table helper == the C ABI's */
  {
    float* a = malloc((size_t)RM_TABLE_FLOATS * 4);
    float* b = malloc((size_t)RM_TABLE_FLOATS * 4);
    struct rm_test_jobject_ ba = buf(a, (size_t)RM_TABLE_FLOATS * 4);
    checks[4] = Java_thi_ng_raymarchcl_Native_makeScatterTable(env, NULL, 1234, &ba) == 0 &&
                rm_make_scatter_table(1234, b) == 0 && memcmp(a, b, (size_t)RM_TABLE_FLOATS * 4) == 0;
    free(a);
    free(b);
  }
  /* device-side gyroid through the shim, copy-out */
  {
    uint8_t* g = malloc(32 * 32 * 32);
    uint8_t* g2 = malloc(32 * 32 * 32);
    struct rm_test_jobject_ bg = buf(g, 32 * 32 * 32);
    checks[5] = Java_thi_ng_raymarchcl_Native_makeGyroidVolume(env, NULL, h, 32, 32, 32, &bg) == 0 &&
                rm_make_gyroid_host(32, 32, 32, g2) == 0;
    int diff = 0;
    for (int i = 0; i < 32 * 32 * 32; i++) diff += g[i] != g2[i];
    checks[5] = checks[5] && diff < 8; /* (device cos/sin may differ from libm in an ulp at a threshold) */
    free(g);
    free(g2);
  }
  /* (the gyroid above replaced the resident volume) */
  g_throws = 0;
  if (Java_thi_ng_raymarchcl_Native_setVolume(env, NULL, h, &bvox, rx, ry, rz) != 0) return 5;
  /* a null REQUIRED buffer -> IllegalArgumentException (no silent error code), optional outputs may be null */
  g_throws = 0;
  Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, NULL, &bmc, iter, n, &bpx, &bargb);
  checks[6] = g_throws == 1 && strcmp(g_thrown_class, "java/lang/IllegalArgumentException") == 0;
  g_throws = 0;
  checks[6] = checks[6] && Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, &bopts, &bmc, iter, n, NULL, &bargb) == 0 &&
              g_throws == 0;
  /* .vox through the shim: written by the C ABI, header and bytes read back by voxInfo / voxLoad */
  {
    char path[512];
    snprintf(path, sizeof path, "%s.vox", argv[2]);
    int32_t res3[3] = {0, 0, 0};
    uint8_t* back = malloc(nvox);
    struct rm_test_jobject_ spath = {path, (jlong)strlen(path), NULL}, bres = buf(res3, 12), bback = buf(back, nvox);
    g_throws = 0;
    checks[7] = rm_vox_save(path, rx, ry, rz, vox) == 0 &&
                Java_thi_ng_raymarchcl_Native_voxInfo(env, NULL, &spath, &bres) == 0 && res3[0] == rx && res3[1] == ry &&
                res3[2] == rz && Java_thi_ng_raymarchcl_Native_voxLoad(env, NULL, &spath, &bback, (jlong)nvox) == 0 &&
                memcmp(back, vox, nvox) == 0 && g_throws == 0 && g_strings_out == 0;
    /* a missing file -> RuntimeException with the library's message */
    struct rm_test_jobject_ nopath = {"/nonexistent/x.vox", 18, NULL};
    Java_thi_ng_raymarchcl_Native_voxInfo(env, NULL, &nopath, &bres);
    checks[7] = checks[7] && g_throws == 1 && strcmp(g_thrown_class, "java/lang/RuntimeException") == 0;
    remove(path);
    free(back);
  }
  /* page-locked caller buffers: the same frame through pinned buffers, then unpinned again */
  g_throws = 0;
  {
    float* pxp = calloc((size_t)n * 4, 4);
    struct rm_test_jobject_ bpp = buf(pxp, (size_t)n * 16);
    if (!checks[6]) fprintf(stderr, "check 6: null-buffer handling failed (%d throws, %s)\n", g_throws, g_thrown_class);
    int ok = 1, step = 0;
#define STEP(x) do { step++; if (ok && !(x)) { ok = 0; fprintf(stderr, "check 6: pin step %d failed: %s\n", step, g_thrown_msg); } } while (0)
    STEP(Java_thi_ng_raymarchcl_Native_pin(env, NULL, h, &bpp, (jlong)n * 16) == 0);
    STEP(Java_thi_ng_raymarchcl_Native_pin(env, NULL, h, &bmc, (jlong)nmc * 4) == 0);
    STEP(Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, &bopts, &bmc, iter, n, &bpp, NULL) == 0);
    STEP(memcmp(pxp, px, (size_t)n * 16) == 0);
    STEP(Java_thi_ng_raymarchcl_Native_unpin(env, NULL, h, &bpp) == 0);
    STEP(Java_thi_ng_raymarchcl_Native_unpin(env, NULL, h, &bmc) == 0);
    STEP(g_throws == 0);
#undef STEP
    checks[6] = checks[6] && ok;
    free(pxp);
  }
  /* staged volumes through the shim: another volume is resident, the scene's volume is staged + committed, the frame
   * equals the first one; a commit without a stage raises */
  {
    float* pxs = calloc((size_t)n * 4, 4);
    struct rm_test_jobject_ bps = buf(pxs, (size_t)n * 16);
    g_throws = 0;
    checks[9] = Java_thi_ng_raymarchcl_Native_makeGyroidVolume(env, NULL, h, rx, ry, rz, NULL) == 0 &&
                Java_thi_ng_raymarchcl_Native_stageVolume(env, NULL, h, &bvox, rx, ry, rz, 32) == 0 &&
                Java_thi_ng_raymarchcl_Native_commitStagedVolume(env, NULL, h) == 0 &&
                Java_thi_ng_raymarchcl_Native_renderFrame(env, NULL, h, &bopts, &bmc, iter, n, &bps, NULL) == 0 &&
                memcmp(pxs, px, (size_t)n * 16) == 0 && g_throws == 0;
    Java_thi_ng_raymarchcl_Native_commitStagedVolume(env, NULL, h);
    checks[9] = checks[9] && g_throws == 1;
    g_throws = 0;
    free(pxs);
  }
  /* the arithmetic contract through the shim: accepted values, a bad value raises */
  g_throws = 0;
  checks[8] = Java_thi_ng_raymarchcl_Native_setContract(env, NULL, h, 1) == 0 &&
              Java_thi_ng_raymarchcl_Native_setContract(env, NULL, h, 2) == 0 &&
              Java_thi_ng_raymarchcl_Native_setContract(env, NULL, h, 0) == 0 && g_throws == 0;
  Java_thi_ng_raymarchcl_Native_setContract(env, NULL, h, 7);
  checks[8] = checks[8] && g_throws == 1;
  Java_thi_ng_raymarchcl_Native_destroy(env, NULL, h);
  f = fopen(argv[2], "wb");
  if (!f) return 9;
  fwrite(px, 4, (size_t)n * 4, f);
  fwrite(argb, 4, n, f);
  fwrite(px1, 4, (size_t)n * 4, f);
  fwrite(argb1, 4, n, f);
  fwrite(checks, 4, 10, f);
  fclose(f);
  return 0;
}
