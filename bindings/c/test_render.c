/*
 * test_render.c -- `test-render` of the reference (core.clj:154-179: build the render state, run the
 * pipeline once, export the image) written against the C ABI alone, the way a native host would use the
 * library: no Python, no torch, only include/raymarch_hip.h and libraymarch_hip.so.
 *
 *   test_render WIDTH HEIGHT ITER VRES MAT OUT.ppm [VOLUME.vox] [MC_SEED]
 *
 * core.clj step                                   | here
 * ------------------------------------------------+---------------------------------------------------
 * cl/init-state, program build (:121-128)         | rm_create
 * vio/load-volume -> v-buf (:146, io.clj:19-33)   | rm_vox_info / rm_vox_load, or make-gyroid-volume
 *                                                 | (generators.clj:27-42) = rm_make_gyroid_host; rm_set_volume
 * compute-eyepos (:150-152), render-options       | rm_compute_eyepos, rm_render_options per pass with
 *   per pass, t = i * 0.333 (:99-106)             | t = i * 0.333
 * generate-scatter-offsets per pass (:134-136)    | rm_make_scatter_table(MC_SEED + i) (the reference seeds from nanoTime)
 * execute-pipeline (:171)                         | rm_render_frame (ARGB only, as the pipeline reads back q-buf)
 * save image (:172-178)                           | binary PPM
 *
 * Prints the FNV-1a hash of the ARGB words (tests/test_c_client.py compares it with the ctypes path).
 * Every failure ends with the library's message on stderr and exit status 2; there is no CPU fallback.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raymarch_hip.h"

static int die(const char* what, int rc) {
  fprintf(stderr, "test_render: %s failed (%d): %s\n", what, rc, rm_last_error());
  return 2;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s WIDTH HEIGHT ITER VRES MAT OUT.ppm [VOLUME.vox] [MC_SEED]\n", argv[0]);
    return 1;
  }
  const int width = atoi(argv[1]), height = atoi(argv[2]), iter = atoi(argv[3]);
  int rx = atoi(argv[4]), ry = rx, rz = rx;
  const char* mat = argv[5];
  const char* out_path = argv[6];
  const char* vox_path = argc > 7 && argv[7][0] ? argv[7] : NULL;
  const uint64_t mc_seed = argc > 8 ? strtoull(argv[8], NULL, 10) : 1000u;
  if (width <= 0 || height <= 0 || iter <= 0 || rx <= 0) {
    fprintf(stderr, "test_render: WIDTH, HEIGHT, ITER and VRES must be positive\n");
    return 1;
  }
  const int n = width * height;
  int rc;

  /* the volume: a .vox file, or the sliced gyroid */
  if (vox_path && (rc = rm_vox_info(vox_path, &rx, &ry, &rz)) != RM_OK) return die("rm_vox_info", rc);
  const size_t cells = (size_t)rx * ry * rz;
  uint8_t* vox = (uint8_t*)malloc(cells);
  void* opts = malloc((size_t)iter * RM_OPTS_BYTES);
  float* mc = (float*)malloc((size_t)iter * RM_TABLE_FLOATS * sizeof(float));
  uint32_t* argb = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
  if (!vox || !opts || !mc || !argb) {
    fprintf(stderr, "test_render: out of memory\n");
    return 2;
  }
  if (vox_path) rc = rm_vox_load(vox_path, vox, cells);
  else rc = rm_make_gyroid_host(rx, ry, rz, vox);
  if (rc != RM_OK) return die(vox_path ? "rm_vox_load" : "rm_make_gyroid_host", rc);

  /* one record and one scatter table per pass */
  rm_render_args a;
  memset(&a, 0, sizeof a);
  a.width = width;
  a.height = height;
  a.vres[0] = rx; a.vres[1] = ry; a.vres[2] = rz;
  a.iter = iter;
  a.fov_deg = a.dof = a.gamma = a.ground_y = a.voxel_size = NAN; /* not given: render-options' defaults */
  if ((rc = rm_compute_eyepos(135.0, 2.25, 0.35, a.eyepos)) != RM_OK) return die("rm_compute_eyepos", rc);
  a.targetpos[0] = 0.0; a.targetpos[1] = -0.4; a.targetpos[2] = 0.0;
  a.mat = mat;
  for (int i = 0; i < iter; i++) {
    a.t = i * 0.333;
    if ((rc = rm_render_options(&a, (char*)opts + (size_t)i * RM_OPTS_BYTES)) != RM_OK) return die("rm_render_options", rc);
    if ((rc = rm_make_scatter_table(mc_seed + (uint64_t)i, mc + (size_t)i * RM_TABLE_FLOATS)) != RM_OK)
      return die("rm_make_scatter_table", rc);
  }

  /* the device: fails here on a box without a gfx950 GPU */
  rm_ctx* ctx = NULL;
  if ((rc = rm_create(0, &ctx)) != RM_OK) return die("rm_create", rc);
  if ((rc = rm_set_volume(ctx, vox, rx, ry, rz)) != RM_OK) return die("rm_set_volume", rc);
  if ((rc = rm_render_frame(ctx, opts, mc, iter, n, NULL, argb)) != RM_OK) return die("rm_render_frame", rc);
  float ms = 0.f;
  int launches = 0;
  (void)rm_last_frame_timing(ctx, &ms, &launches);
  rm_destroy(ctx);

  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; i++)
    for (int b = 0; b < 4; b++) {
      h ^= (argb[i] >> (8 * b)) & 0xffu;
      h *= 1099511628211ull;
    }
  FILE* f = fopen(out_path, "wb");
  if (!f) {
    fprintf(stderr, "test_render: cannot write %s\n", out_path);
    return 2;
  }
  fprintf(f, "P6\n%d %d\n255\n", width, height);
  for (int i = 0; i < n; i++) {
    const unsigned char rgb[3] = {(unsigned char)(argb[i] >> 16), (unsigned char)(argb[i] >> 8), (unsigned char)argb[i]};
    fwrite(rgb, 1, 3, f);
  }
  fclose(f);
  printf("%dx%d x %d passes, %dx%dx%d volume, %s: argb fnv1a64 %016llx, device time %.3f ms in %d launch(es) -> %s\n",
         width, height, iter, rx, ry, rz, mat, (unsigned long long)h, ms, launches, out_path);
  free(vox); free(opts); free(mc); free(argb);
  return 0;
}
