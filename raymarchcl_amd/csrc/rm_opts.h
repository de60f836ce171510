// rm_opts.h -- the 544-byte per-pass option record, as plain C++ PODs.
//
// Replaces the reference's TRenderOpts / TMaterial typedefs
// (/root/reference/resources/renderer.cl:14-19, 35-78), which the Clojure host
// encodes with thi.ng/structgen (core.clj:25-26, 99-106).  OpenCL alignment:
// float3 occupies 16 bytes (4th lane is padding); little-endian.
#pragma once
#include <cstddef>
#include <cstdint>

struct alignas(16) RmMaterial {
  float albedo[4];
  float r0;
  float smoothness;
  float dummy[2];
};

struct alignas(16) RmOpts {
  float eyePos[4];
  float targetPos[4];
  float up[4];
  float voxelBounds[4];
  float voxelBounds2[4];
  float voxelBoundsMin[4];
  float voxelBoundsMax[4];
  float invVoxelScale[4];
  float skyColor1[4];
  float skyColor2[4];
  int32_t voxelRes[4];  // x, y, z, x*y
  int32_t resolution[2];
  float invAspect;
  float time;
  float fov;
  int32_t maxIter;
  int32_t maxVoxelIter;
  float maxDist;
  float startDist;
  float eps;
  int32_t aoIter;
  float aoStepDist;
  float aoAmp;
  float voxelSize;
  float groundY;
  int32_t shadowIter;
  int32_t reflectIter;
  float shadowBias;
  float lightScatter;
  float minLightAtt;
  float gamma;
  float exposure;
  float dof;
  float frameBlend;
  float fogPow;
  float flareAmp;
  int32_t mcTableLength;
  uint8_t isoVal;
  uint8_t numLights;
  uint8_t pad_[2];
  float lightPos[4][4];
  float lightColor[4][4];
  RmMaterial materials[4];
};

// AO probes (aoIter + 1) whose results a wavefront's exchange area holds per lane (rm_shade.hpp occlusion_wave): frames whose
// records ask for more go through the single-pass kernels (rm_api.hip frame_on_device), one launch per pass
#define RM_WAVE_AO_PROBES 8
#define RM_OPTS_SIZE 544
#define RM_TABLE_ENTRIES 0x4000
static_assert(sizeof(RmMaterial) == 32, "TMaterial is 32 bytes");
static_assert(sizeof(RmOpts) == RM_OPTS_SIZE, "TRenderOpts is 544 bytes");
static_assert(offsetof(RmOpts, targetPos) == 16 && offsetof(RmOpts, up) == 32, "layout");
static_assert(offsetof(RmOpts, voxelBounds) == 48 && offsetof(RmOpts, voxelBounds2) == 64, "layout");
static_assert(offsetof(RmOpts, voxelBoundsMin) == 80 && offsetof(RmOpts, voxelBoundsMax) == 96, "layout");
static_assert(offsetof(RmOpts, invVoxelScale) == 112 && offsetof(RmOpts, skyColor1) == 128, "layout");
static_assert(offsetof(RmOpts, skyColor2) == 144 && offsetof(RmOpts, voxelRes) == 160, "layout");
static_assert(offsetof(RmOpts, resolution) == 176 && offsetof(RmOpts, invAspect) == 184, "layout");
static_assert(offsetof(RmOpts, time) == 188 && offsetof(RmOpts, fov) == 192, "layout");
static_assert(offsetof(RmOpts, maxIter) == 196 && offsetof(RmOpts, maxVoxelIter) == 200, "layout");
static_assert(offsetof(RmOpts, maxDist) == 204 && offsetof(RmOpts, startDist) == 208, "layout");
static_assert(offsetof(RmOpts, eps) == 212 && offsetof(RmOpts, aoIter) == 216, "layout");
static_assert(offsetof(RmOpts, aoStepDist) == 220 && offsetof(RmOpts, aoAmp) == 224, "layout");
static_assert(offsetof(RmOpts, voxelSize) == 228 && offsetof(RmOpts, groundY) == 232, "layout");
static_assert(offsetof(RmOpts, shadowIter) == 236 && offsetof(RmOpts, reflectIter) == 240, "layout");
static_assert(offsetof(RmOpts, shadowBias) == 244 && offsetof(RmOpts, lightScatter) == 248, "layout");
static_assert(offsetof(RmOpts, minLightAtt) == 252 && offsetof(RmOpts, gamma) == 256, "layout");
static_assert(offsetof(RmOpts, exposure) == 260 && offsetof(RmOpts, dof) == 264, "layout");
static_assert(offsetof(RmOpts, frameBlend) == 268 && offsetof(RmOpts, fogPow) == 272, "layout");
static_assert(offsetof(RmOpts, flareAmp) == 276 && offsetof(RmOpts, mcTableLength) == 280, "layout");
static_assert(offsetof(RmOpts, isoVal) == 284 && offsetof(RmOpts, numLights) == 285, "layout");
static_assert(offsetof(RmOpts, lightPos) == 288 && offsetof(RmOpts, lightColor) == 352, "layout");
static_assert(offsetof(RmOpts, materials) == 416, "layout");
