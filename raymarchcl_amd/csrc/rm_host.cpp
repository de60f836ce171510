// rm_host.cpp -- host-side parameter layer of the C ABI (no device code): what a
// non-Python host (the JNI shim, a C++ tool) needs to drive the render path without
// re-implementing the reference's Clojure helpers.
//
//   rm_render_options     render-options      core.clj:28-74  (+ presets, materials.clj:3-76)
//   rm_compute_eyepos     compute-eyepos      core.clj:150-152
//   rm_make_scatter_table generate-scatter-offsets  generators.clj:8-16 (seeded, see below)
//   rm_make_gyroid_host   make-gyroid-volume  generators.clj:27-42
//   rm_vox_save / rm_vox_info / rm_vox_load   save-volume / load-volume  io.clj:9-33
//
// Byte-for-byte the same results as the Python host layer (raymarchcl_amd/options.py,
// structs.py, generators.py, vio.py); tests/test_host_abi.py compares them.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/raymarch_hip.h"
#include "rm_opts.h"

extern "C" int rm_host_fail_(int code, const char* msg);  // rm_api.hip: sets rm_last_error()

namespace {

struct Mat { double albedo[4]; double r0, smoothness; };
struct Preset {
  const char* name;
  int n_colors;
  double lightColor[2][4];
  bool has_pos;
  double lightPos[2][4];
  Mat mats[4];
  int numLights;
  double aoAmp;
  int reflectIter;
};

// materials.clj:3-76
const Preset kPresets[] = {
    {"orange-stripes", 2, {{28, 18, 8, 0}, {8, 18, 28, 0}}, true, {{-2, 0, -2, 0}, {2, 0, 2, 0}},
     {{{1.0, 1.0, 1.0, 1.0}, 0.1, 0.9}, {{4.9, 0.9, 0.05, 1.0}, 0.01, 0.5},
      {{1.9, 1.9, 1.9, 1.0}, 0.01, 0.4}, {{0.9, 0.9, 0.9, 1.0}, 0.8, 0.1}}, 2, 0.25, 1},
    {"metal", 2, {{28, 18, 8, 0}, {16, 36, 56, 0}}, true, {{0, 2, 0, 0}, {3, 0, 3, 0}},
     {{{0.01, 0.01, 0.01, 1.0}, 0.1, 0.5}, {{1.9, 1.9, 1.9, 1.0}, 0.1, 0.5},
      {{0.25, 0.27, 0.5, 1.0}, 0.7, 0.1}, {{1.0, 1.0, 1.0, 1.0}, 0.2, 0.1}}, 2, 0.25, 3},
    {"metal2", 2, {{28, 18, 8, 0}, {8, 18, 28, 0}}, true, {{-2, 0, -2, 0}, {2, 0, 2, 0}},
     {{{0.0, 0.0, 0.0, 1.0}, 0.1, 0.9}, {{1.0, 1.01, 1.075, 1.0}, 0.4, 0.7},
      {{1.9, 1.9, 1.9, 1.0}, 0.4, 0.5}, {{0.9, 0.9, 0.9, 1.0}, 0.75, 0.2}}, 2, 0.25, 3},
    {"ao", 1, {{50, 50, 50, 0}, {0, 0, 0, 0}}, false, {{0, 0, 0, 0}, {0, 0, 0, 0}},
     {{{1, 1, 1, 1}, 0.0, 1.0}, {{1, 1, 1, 1}, 0.0, 1.0}, {{1, 1, 1, 1}, 0.0, 1.0}, {{1, 1, 1, 1}, 0.0, 1.0}},
     1, 0.25, 0},
};

const Preset& preset_of(const char* mat) {
  if (mat && mat[0] == ':') mat++;
  if (mat)
    for (const Preset& p : kPresets)
      if (strcmp(p.name, mat) == 0) return p;
  return kPresets[3];  // (get presets mat (presets :ao)), core.clj:74
}

inline bool given(double v) { return !std::isnan(v); }
inline void set3(float* dst, double x, double y, double z) {
  dst[0] = (float)x; dst[1] = (float)y; dst[2] = (float)z; dst[3] = 0.0f;
}

const double kRad = 3.14159265358979323846 / 180.0;  // thi.ng.math RAD

uint64_t splitmix(uint64_t seed, uint64_t index1) {  // index1 = 1, 2, ... (generators.py _splitmix64)
  uint64_t z = seed + index1 * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

}  // namespace

extern "C" {

int rm_compute_eyepos(double theta_deg, double dist, double y, double out_xyz[3]) {
  if (!out_xyz) return rm_host_fail_(RM_EINVAL, "out_xyz is NULL");
  const double t = theta_deg * kRad, s = std::sin(t), c = std::cos(t);
  const double x = 0.0, z = dist;  // (g/rotate-y (vec3 0 y dist) theta)
  out_xyz[0] = x * c + z * s;
  out_xyz[1] = y;
  out_xyz[2] = z * c - x * s;
  return RM_OK;
}

int rm_render_options(const rm_render_args* a, void* out544) {
  if (!a || !out544) return rm_host_fail_(RM_EINVAL, "NULL argument");
  if (a->width <= 0 || a->height <= 0 || a->iter <= 0 || a->vres[0] <= 0 || a->vres[1] <= 0 ||
      a->vres[2] <= 0)
    return rm_host_fail_(RM_EINVAL, "width/height/iter/vres must be positive");
  RmOpts o;
  memset(&o, 0, sizeof o);
  const double clip = 0.99;
  if (given(a->eyepos[0])) set3(o.eyePos, a->eyepos[0], a->eyepos[1], a->eyepos[2]);
  else set3(o.eyePos, 2, 0, 2);
  if (given(a->targetpos[0])) set3(o.targetPos, a->targetpos[0], a->targetpos[1], a->targetpos[2]);
  else set3(o.targetPos, 0, -0.15, 0);
  set3(o.up, 0, 1, 0);
  set3(o.voxelBounds, 1, 1, 1);
  set3(o.voxelBounds2, 2, 2, 2);
  set3(o.voxelBoundsMin, -clip, -clip, -clip);
  set3(o.voxelBoundsMax, clip, clip, clip);
  set3(o.invVoxelScale, 0.5, 0.5, 0.5);
  set3(o.skyColor1, 1.8, 1.8, 1.9);
  set3(o.skyColor2, 0.1, 0.1, 0.1);
  o.voxelRes[0] = a->vres[0]; o.voxelRes[1] = a->vres[1]; o.voxelRes[2] = a->vres[2];
  o.voxelRes[3] = a->vres[0] * a->vres[1];
  o.resolution[0] = a->width; o.resolution[1] = a->height;
  o.invAspect = (float)((double)a->height / (double)a->width);
  o.time = (float)a->t;
  o.fov = (float)((given(a->fov_deg) ? a->fov_deg : 90.0) * kRad);
  o.maxIter = 128; o.maxVoxelIter = 192;
  o.maxDist = 30.0f; o.startDist = 0.0f; o.eps = (float)0.005;
  o.aoIter = 5; o.aoStepDist = (float)0.05;
  o.voxelSize = (float)(given(a->voxel_size) ? a->voxel_size : 1.0 / (double)a->vres[0]);
  o.groundY = (float)(given(a->ground_y) ? a->ground_y : 1.05);
  o.shadowIter = 128;
  o.shadowBias = (float)0.1; o.lightScatter = (float)0.2; o.minLightAtt = 0.0f;
  o.gamma = (float)(given(a->gamma) ? a->gamma : 1.5);
  o.exposure = 3.5f;
  o.dof = (float)(given(a->dof) ? a->dof : 0.001);
  o.frameBlend = (float)(1.0 / (double)a->iter);
  o.fogPow = (float)0.05; o.flareAmp = (float)0.015;
  o.isoVal = 32;
  // defaults that the preset (merged last, core.clj:74) may replace
  const double defPos[2][4] = {{-2, 0, -2, 0}, {2, 0, 2, 0}};
  const Preset& p = preset_of(a->mat);
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 4; k++) o.lightPos[i][k] = (float)(p.has_pos ? p.lightPos[i][k] : defPos[i][k]);
  for (int i = 0; i < p.n_colors; i++)
    for (int k = 0; k < 4; k++) o.lightColor[i][k] = (float)p.lightColor[i][k];
  for (int i = 0; i < 4; i++) {
    for (int k = 0; k < 4; k++) o.materials[i].albedo[k] = (float)p.mats[i].albedo[k];
    o.materials[i].r0 = (float)p.mats[i].r0;
    o.materials[i].smoothness = (float)p.mats[i].smoothness;
  }
  o.numLights = (uint8_t)p.numLights;
  o.aoAmp = (float)p.aoAmp;
  o.reflectIter = p.reflectIter;
  memcpy(out544, &o, sizeof o);
  return RM_OK;
}

int rm_make_scatter_table(uint64_t seed, float* out) {
  if (!out) return rm_host_fail_(RM_EINVAL, "out is NULL");
  for (uint64_t i = 0; i < 0x4000; i++) {
    float v[4];
    double s2 = 0.0;
    for (int k = 0; k < 4; k++) {
      const double u = (double)(splitmix(seed, 4 * i + k + 1) >> 11) * (1.0 / 9007199254740992.0);
      v[k] = (float)(2.0 * u - 1.0);  // (float (- (* 2.0 nextDouble) 1.0))
      s2 += (double)v[k] * (double)v[k];
    }
    const double m = 1.0 / std::sqrt(s2);
    for (int k = 0; k < 4; k++) out[4 * i + k] = (float)((double)v[k] * m);
  }
  return RM_OK;
}

int rm_make_gyroid_host(int rx, int ry, int rz, uint8_t* out) {
  if (!out || rx <= 0 || ry <= 0 || rz <= 0) return rm_host_fail_(RM_EINVAL, "bad argument");
  const double scl = 0.01 * (512.0 / (double)rx);
  memset(out, 0, (size_t)rx * ry * rz);
  std::vector<double> cx(rx), sx(rx), cy(ry), sy(ry);
  for (int x = 0; x < rx; x++) { const double X = x * scl + 0.3875; cx[x] = std::cos(X); sx[x] = std::sin(X); }
  for (int y = 0; y < ry; y++) { const double Y = y * scl + 0.0; cy[y] = std::cos(Y); sy[y] = std::sin(Y); }
  for (int z = 0; z < rz; z++) {
    if ((z & 0x3f) < 32) continue;
    const double Z = z * scl + 0.0, cz = std::cos(Z), sz = std::sin(Z);
    for (int y = 0; y < ry; y++)
      for (int x = 0; x < rx; x++) {
        const double v = std::fabs(cx[x] * sz + cy[y] * sx[x] + cz * sy[y]) - 1.0;
        uint8_t b = 0;
        if (std::fabs(0.2 - v) < 0.05) b = (x & 0x3f) < 32 ? 64 : 128;
        else if (v > 0.35) b = 255;
        out[((size_t)z * ry + y) * rx + x] = b;
      }
  }
  return RM_OK;
}

// ---- .vox files: "VOXEL", 3 x int32 big-endian, 1 byte element size, raw bytes (io.clj:9-33)
int rm_vox_save(const char* path, int rx, int ry, int rz, const uint8_t* voxels) {
  if (!path || !voxels || rx <= 0 || ry <= 0 || rz <= 0) return rm_host_fail_(RM_EINVAL, "bad argument");
  FILE* f = fopen(path, "wb");
  if (!f) return rm_host_fail_(RM_EINVAL, "cannot open file for writing");
  unsigned char h[18] = {'V', 'O', 'X', 'E', 'L'};
  const int r[3] = {rx, ry, rz};
  for (int i = 0; i < 3; i++)
    for (int b = 0; b < 4; b++) h[5 + 4 * i + b] = (unsigned char)((uint32_t)r[i] >> (24 - 8 * b));
  h[17] = 1;
  const size_t n = (size_t)rx * ry * rz;
  const bool ok = fwrite(h, 1, 18, f) == 18 && fwrite(voxels, 1, n, f) == n;
  fclose(f);
  return ok ? RM_OK : rm_host_fail_(RM_EINVAL, "short write");
}
int rm_vox_info(const char* path, int* rx, int* ry, int* rz) {
  if (!path || !rx || !ry || !rz) return rm_host_fail_(RM_EINVAL, "bad argument");
  FILE* f = fopen(path, "rb");
  if (!f) return rm_host_fail_(RM_EINVAL, "cannot open file");
  unsigned char h[18];
  const bool ok = fread(h, 1, 18, f) == 18 && memcmp(h, "VOXEL", 5) == 0 && h[17] == 1;
  fclose(f);
  if (!ok) return rm_host_fail_(RM_EINVAL, "not a VOXEL volume file");
  int* r[3] = {rx, ry, rz};
  for (int i = 0; i < 3; i++)
    *r[i] = (int)(((uint32_t)h[5 + 4 * i] << 24) | ((uint32_t)h[6 + 4 * i] << 16) | ((uint32_t)h[7 + 4 * i] << 8) |
                  (uint32_t)h[8 + 4 * i]);
  if (*rx <= 0 || *ry <= 0 || *rz <= 0) return rm_host_fail_(RM_EINVAL, "bad resolution in header");
  return RM_OK;
}
int rm_vox_load(const char* path, uint8_t* out, size_t capacity) {
  int rx, ry, rz;
  int rc = rm_vox_info(path, &rx, &ry, &rz);
  if (rc) return rc;
  const size_t n = (size_t)rx * ry * rz;
  if (!out || capacity < n) return rm_host_fail_(RM_EINVAL, "output buffer too small");
  FILE* f = fopen(path, "rb");
  if (!f) return rm_host_fail_(RM_EINVAL, "cannot open file");
  const bool ok = fseek(f, 18, SEEK_SET) == 0 && fread(out, 1, n, f) == n;
  fclose(f);
  return ok ? RM_OK : rm_host_fail_(RM_EINVAL, "truncated volume");
}

}  // extern "C"
