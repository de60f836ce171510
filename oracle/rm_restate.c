/*
 * oracle/rm_restate.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement (plain C, scalar float32) of the reference's render path:
 * the two OpenCL kernels RenderImage / TonemapImage and their 28 device
 * functions in /root/reference/resources/renderer.cl, plus the pass sequence
 * of make-pipeline (/root/reference/src/thi/ng/raymarchcl/core.clj:76-97).
 * Written from the behaviour of that source (each function cites the lines it
 * follows), not copied from it: own data structures, scalar arithmetic, raw
 * byte access to the 544-byte option record.
 *
 * Purpose: (1) the parity checker for the HIP path in tests/ and smoke(),
 * (2) the "cpu_baseline" leg of bench.py, (3) the counter of algorithmic
 * bytes (in-bounds voxel reads, scatter-table reads) that the roofline figure
 * is computed from.  The product path never links or calls this file.
 *
 * Pinning: in the build container this file is checked bit-for-bit against
 * oracle/_ref/libref_oracle.so (the unmodified reference kernel compiled for
 * x86-64, see oracle/Makefile and tests/test_oracle_vs_reference.py) and
 * against the fixtures in tests/golden/ that were generated from that build.
 *
 * Arithmetic contract (shared with the HIP kernels): IEEE-754 binary32,
 * round-to-nearest-even, NO fused multiply-add, operations in the order the
 * reference source writes them; built-ins as defined in oracle/cl_scalar.h.
 * Build with -ffp-contract=off.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "cl_scalar.h"

typedef struct { float x, y, z; } v3;

/* ---- option record: 544 bytes, layout of renderer.cl:35-78 ---- */
enum {
  O_eyePos = 0, O_targetPos = 16, O_up = 32, O_voxelBounds = 48, O_voxelBounds2 = 64,
  O_voxelBoundsMin = 80, O_voxelBoundsMax = 96, O_invVoxelScale = 112, O_skyColor1 = 128,
  O_skyColor2 = 144, O_voxelRes = 160, O_resolution = 176, O_invAspect = 184, O_time = 188,
  O_fov = 192, O_maxIter = 196, O_maxVoxelIter = 200, O_maxDist = 204, O_startDist = 208,
  O_eps = 212, O_aoIter = 216, O_aoStepDist = 220, O_aoAmp = 224, O_voxelSize = 228,
  O_groundY = 232, O_shadowIter = 236, O_reflectIter = 240, O_shadowBias = 244,
  O_lightScatter = 248, O_minLightAtt = 252, O_gamma = 256, O_exposure = 260, O_dof = 264,
  O_frameBlend = 268, O_fogPow = 272, O_flareAmp = 276, O_mcTableLength = 280, O_isoVal = 284,
  O_numLights = 285, O_lightPos = 288, O_lightColor = 352, O_materials = 416,
  OPTS_SIZE = 544, MAT_SIZE = 32, MAT_r0 = 16, MAT_smoothness = 20
};

typedef struct {
  uint64_t vox_reads;     /* in-bounds voxel byte loads (voxelLookup + voxelLookupI) */
  uint64_t mc_reads;      /* scatter-table float4 loads */
  uint64_t rays;          /* raymarch() invocations */
  uint64_t dts_calls;     /* distanceToScene() invocations */
  uint64_t march_steps;   /* inner fixed-step samples (voxelLookup calls incl. out-of-bounds) */
  uint64_t ao_calls;      /* ambientOcclusion() invocations */
  uint64_t primary_hits;  /* samples whose primary ray hit something */
  uint64_t oob_material;  /* material index outside the option record: undefined in the reference */
} rmo_stats;

typedef struct {
  const uint8_t* vox;
  const float* mc;
  const float* sdf; /* quality mode (NOT the reference's algorithm): float distance field, or NULL */
  uint8_t raw[OPTS_SIZE];
  /* decoded copies of the hot fields */
  v3 eyePos, targetPos, up, vb, vb2, vbMin, vbMax, ivs, sky1, sky2;
  int rx, ry, rz, rxy, resx, resy;
  float invAspect, time, fov, maxDist, startDist, eps, aoStepDist, aoAmp, voxelSize, groundY;
  float shadowBias, lightScatter, minLightAtt, exposure, dof, frameBlend, fogPow, flareAmp;
  int maxIter, maxVoxelIter, aoIter, shadowIter, reflectIter;
  int isoVal, numLights;
  rmo_stats st;
} ctx_t;

static float ldf(const uint8_t* raw, int off) { float f; memcpy(&f, raw + off, 4); return f; }
static int ldi(const uint8_t* raw, int off) { int32_t i; memcpy(&i, raw + off, 4); return i; }
static v3 ld3(const uint8_t* raw, int off) { v3 v = {ldf(raw, off), ldf(raw, off + 4), ldf(raw, off + 8)}; return v; }

static void ctx_init(ctx_t* c, const uint8_t* vox, const float* mc, const void* opts544) {
  c->sdf = NULL;
  memset(c, 0, sizeof *c);
  c->vox = vox; c->mc = mc;
  memcpy(c->raw, opts544, OPTS_SIZE);
  const uint8_t* r = c->raw;
  c->eyePos = ld3(r, O_eyePos); c->targetPos = ld3(r, O_targetPos); c->up = ld3(r, O_up);
  c->vb = ld3(r, O_voxelBounds); c->vb2 = ld3(r, O_voxelBounds2);
  c->vbMin = ld3(r, O_voxelBoundsMin); c->vbMax = ld3(r, O_voxelBoundsMax);
  c->ivs = ld3(r, O_invVoxelScale); c->sky1 = ld3(r, O_skyColor1); c->sky2 = ld3(r, O_skyColor2);
  c->rx = ldi(r, O_voxelRes); c->ry = ldi(r, O_voxelRes + 4); c->rz = ldi(r, O_voxelRes + 8);
  c->rxy = ldi(r, O_voxelRes + 12);
  c->resx = ldi(r, O_resolution); c->resy = ldi(r, O_resolution + 4);
  c->invAspect = ldf(r, O_invAspect); c->time = ldf(r, O_time); c->fov = ldf(r, O_fov);
  c->maxIter = ldi(r, O_maxIter); c->maxVoxelIter = ldi(r, O_maxVoxelIter);
  c->maxDist = ldf(r, O_maxDist); c->startDist = ldf(r, O_startDist); c->eps = ldf(r, O_eps);
  c->aoIter = ldi(r, O_aoIter); c->aoStepDist = ldf(r, O_aoStepDist); c->aoAmp = ldf(r, O_aoAmp);
  c->voxelSize = ldf(r, O_voxelSize); c->groundY = ldf(r, O_groundY);
  c->shadowIter = ldi(r, O_shadowIter); c->reflectIter = ldi(r, O_reflectIter);
  c->shadowBias = ldf(r, O_shadowBias); c->lightScatter = ldf(r, O_lightScatter);
  c->minLightAtt = ldf(r, O_minLightAtt); c->exposure = ldf(r, O_exposure); c->dof = ldf(r, O_dof);
  c->frameBlend = ldf(r, O_frameBlend); c->fogPow = ldf(r, O_fogPow); c->flareAmp = ldf(r, O_flareAmp);
  c->isoVal = r[O_isoVal]; c->numLights = r[O_numLights];
}

/* materials[id] addressed as bytes: ids whose 32-byte record lies inside the
 * option record (-13..3) read whatever bytes are there, exactly like the
 * reference's private copy would; anything else is undefined in the reference
 * (renderer.cl:394,418,437) -- counted, and defined here as an all-zero material. */
typedef struct { v3 albedo; float r0, smoothness; } mat_t;
static mat_t material(ctx_t* c, int id) {
  mat_t m = {{0, 0, 0}, 0, 0};
  const long off = (long)O_materials + (long)MAT_SIZE * (long)id;
  if (off < 0 || off + MAT_SIZE > OPTS_SIZE) { c->st.oob_material++; return m; }
  m.albedo = ld3(c->raw, (int)off);
  m.r0 = ldf(c->raw, (int)off + MAT_r0);
  m.smoothness = ldf(c->raw, (int)off + MAT_smoothness);
  return m;
}
static v3 light_pos_opt(const ctx_t* c, int i) { return ld3(c->raw, O_lightPos + 16 * i); }
static v3 light_color_opt(const ctx_t* c, int i) { return ld3(c->raw, O_lightColor + 16 * i); }

/* ---- tiny vector helpers: every one is a fixed sequence of IEEE ops ---- */
static inline v3 V(float x, float y, float z) { v3 v = {x, y, z}; return v; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* mad(a,b,c) = a*b + c, unfused */
static inline v3 mad3(v3 a, v3 b, v3 c) { return V(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z); }
static inline v3 mad3s(v3 a, float s, v3 c) { return V(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z); }
static inline v3 mix3s(v3 a, v3 b, float t) {
  return V(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t);
}
static inline v3 normalize3(v3 v) {
  if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
  const float s = 1.0f / cl_sqrt(dot3(v, v));
  return muls(v, s);
}
static inline float length3(v3 v) { return cl_sqrt(dot3(v, v)); }

/* ---- scatter table: renderer.cl:142-144 ---- */
typedef struct { float x, y, z, w; } v4;
static v4 table(ctx_t* c, uint32_t seed) {
  const float* p = c->mc + 4u * (seed & 0x3fffu);
  v4 r = {p[0], p[1], p[2], p[3]};
  c->st.mc_reads++;
  return r;
}

/* ---- slab test against the clip box: renderer.cl:153-161 ---- */
static float box_entry(const ctx_t* c, v3 p, v3 d) {
  const v3 lo = V((c->vbMin.x - p.x) / d.x, (c->vbMin.y - p.y) / d.y, (c->vbMin.z - p.z) / d.z);
  const v3 hi = V((c->vbMax.x - p.x) / d.x, (c->vbMax.y - p.y) / d.y, (c->vbMax.z - p.z) / d.z);
  const float nx = cl_min(hi.x, lo.x), ny = cl_min(hi.y, lo.y), nz = cl_min(hi.z, lo.z);
  const float a = cl_max(cl_max(nx, 0.0f), cl_max(ny, nz));
  const float fx = cl_max(hi.x, lo.x), fy = cl_max(hi.y, lo.y), fz = cl_max(hi.z, lo.z);
  const float b = cl_min(fx, cl_min(fy, fz));
  return b > a ? a : -1.0f;
}

/* ---- nearest-voxel fetch at a normalised position: renderer.cl:163-170 ---- */
static inline int in_grid(const ctx_t* c, int qx, int qy, int qz) {
  return qz >= 0 && qz < c->rz && qy >= 0 && qy < c->ry && qx >= 0 && qx < c->rx;
}
static inline void cell_of(const ctx_t* c, v3 p, int* qx, int* qy, int* qz) {
  *qx = cl_convert_int_sat(p.x * (float)c->rx);
  *qy = cl_convert_int_sat(p.y * (float)c->ry);
  *qz = cl_convert_int_sat(p.z * (float)c->rz);
}
static int voxel_at(ctx_t* c, v3 p) {
  int qx, qy, qz;
  cell_of(c, p, &qx, &qy, &qz);
  c->st.march_steps++;
  if (!in_grid(c, qx, qy, qz)) return -1;
  c->st.vox_reads++;
  return (int)c->vox[qz * c->rxy + qy * c->rx + qx];
}
/* binary occupancy at an integer cell: renderer.cl:172-178 (>= isoVal, 0 outside) */
static float solid(ctx_t* c, int qx, int qy, int qz) {
  if (!in_grid(c, qx, qy, qz)) return 0.0f;
  c->st.vox_reads++;
  return cl_step((float)c->isoVal, (float)c->vox[qz * c->rxy + qy * c->rx + qx]);
}
/* negated central difference of occupancy: renderer.cl:180-188 */
static v3 cell_gradient(ctx_t* c, int qx, int qy, int qz) {
  const float gx = solid(c, qx + 1, qy, qz) - solid(c, qx - 1, qy, qz);
  const float gy = solid(c, qx, qy + 1, qz) - solid(c, qx, qy - 1, qz);
  const float gz = solid(c, qx, qy, qz + 1) - solid(c, qx, qy, qz - 1);
  return V(-gx, -gy, -gz);
}
/* sum of the gradients of the solid cells in the 3x3x3 block, normalised: renderer.cl:190-203 */
static v3 smooth_gradient(ctx_t* c, int qx, int qy, int qz) {
  v3 n = V(0.0f, 0.0f, 0.0f);
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++)
        if (solid(c, qx + dx, qy + dy, qz + dz) > 0.0f)
          n = add(n, cell_gradient(c, qx + dx, qy + dy, qz + dz));
  return normalize3(n);
}
/* value band -> material slot: renderer.cl:205-207 */
static float band(int v) { return v < 168 ? (v < 84 ? 1.0f : 2.0f) : 3.0f; }

/* ---- distance estimate at rpos along dir: renderer.cl:209-237 ----
 * returns distance in *dist, material code in *code, surface normal in *nrm */
/* ======================================================================================
 * QUALITY MODE -- not a reference feature and not reference-equivalent:
 * the estimate comes from a float distance field sampled trilinearly, the hit normal is
 * its gradient, shadows are soft.  Everything around it (march = sphere tracing with the
 * estimate as the step, AO, lighting, reflections, atmosphere, blend, tonemap) is the code
 * above.  Same arithmetic contract, so the HIP kernel can be checked bit for bit.
 * ====================================================================================== */
/* field value at a position inside (or on) the clip box: trilinear over cell centres */
static float sdf_sample(const ctx_t* c, v3 q) {
  const float ux = cl_clamp((q.x + c->vb.x) * c->ivs.x * (float)c->rx - 0.5f, 0.0f, (float)(c->rx - 1));
  const float uy = cl_clamp((q.y + c->vb.y) * c->ivs.y * (float)c->ry - 0.5f, 0.0f, (float)(c->ry - 1));
  const float uz = cl_clamp((q.z + c->vb.z) * c->ivs.z * (float)c->rz - 0.5f, 0.0f, (float)(c->rz - 1));
  int ix = (int)ux, iy = (int)uy, iz = (int)uz; /* u >= 0: truncation = floor */
  if (ix > c->rx - 2) ix = c->rx - 2;
  if (iy > c->ry - 2) iy = c->ry - 2;
  if (iz > c->rz - 2) iz = c->rz - 2;
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (iz < 0) iz = 0;
  const float fx = ux - (float)ix, fy = uy - (float)iy, fz = uz - (float)iz;
  const int x1 = ix + 1 < c->rx ? ix + 1 : ix, y1 = iy + 1 < c->ry ? iy + 1 : iy,
            z1 = iz + 1 < c->rz ? iz + 1 : iz;
  const float* g = c->sdf;
#define SDF_AT(X, Y, Z) g[((size_t)(Z) * c->ry + (Y)) * c->rx + (X)]
  const float a00 = SDF_AT(ix, iy, iz) + (SDF_AT(x1, iy, iz) - SDF_AT(ix, iy, iz)) * fx;
  const float a10 = SDF_AT(ix, y1, iz) + (SDF_AT(x1, y1, iz) - SDF_AT(ix, y1, iz)) * fx;
  const float a01 = SDF_AT(ix, iy, z1) + (SDF_AT(x1, iy, z1) - SDF_AT(ix, iy, z1)) * fx;
  const float a11 = SDF_AT(ix, y1, z1) + (SDF_AT(x1, y1, z1) - SDF_AT(ix, y1, z1)) * fx;
#undef SDF_AT
  const float b0 = a00 + (a10 - a00) * fy;
  const float b1 = a01 + (a11 - a01) * fy;
  return b0 + (b1 - b0) * fz;
}
/* distance to the field's zero set from anywhere: outside the clip box the distance to the
 * box is added to the value at the nearest point of the box */
static float sdf_volume(const ctx_t* c, v3 p) {
  const v3 q = V(cl_clamp(p.x, c->vbMin.x, c->vbMax.x), cl_clamp(p.y, c->vbMin.y, c->vbMax.y),
                 cl_clamp(p.z, c->vbMin.z, c->vbMax.z));
  return sdf_sample(c, q) + length3(sub(p, q));
}
static void scene_distance_sdf(ctx_t* c, v3 rpos, v3 dir, float* dist, float* code, v3* nrm) {
  const float h = rpos.y + c->groundY;
  float rd, rc;
  if (h < 1e5f) { rd = h; rc = h; } else { rd = 1e5f; rc = -1.0f; }
  *nrm = ((double)rd < 1e5) ? V(0.0f, 1.0f, 0.0f) : neg(dir);
  const float dv = sdf_volume(c, rpos);
  if (dv < rd) {
    rd = dv;
    rc = 1.0f;
    if (dv <= c->eps * 2.0f) { /* close enough to be the hit: gradient by central differences */
      const float e = c->voxelSize;
      const v3 g = V(sdf_volume(c, V(rpos.x + e, rpos.y, rpos.z)) - sdf_volume(c, V(rpos.x - e, rpos.y, rpos.z)),
                     sdf_volume(c, V(rpos.x, rpos.y + e, rpos.z)) - sdf_volume(c, V(rpos.x, rpos.y - e, rpos.z)),
                     sdf_volume(c, V(rpos.x, rpos.y, rpos.z + e)) - sdf_volume(c, V(rpos.x, rpos.y, rpos.z - e)));
      *nrm = normalize3(g);
    } else {
      *nrm = neg(dir);
    }
  }
  *dist = rd;
  *code = rc;
}
/* penumbra estimate along a light ray: min over the march of k * clearance / distance */
static float scene_only_distance_sdf(ctx_t* c, v3 p) {
  const float h = p.y + c->groundY;
  const float g = h < 1e5f ? h : 1e5f;
  return cl_min(g, sdf_volume(c, p));
}
static float soft_shadow_sdf(ctx_t* c, v3 p, v3 ldir, float lmax) {
  const float k = 1.0f / cl_max(c->lightScatter, 0.01f);
  float res = 1.0f;
  float t = 0.0f;
  for (int i = 0; i < c->shadowIter; i++) {
    const float d = scene_only_distance_sdf(c, mad3s(ldir, t, p));
    if (d <= c->eps * 0.5f) return 0.0f;
    res = cl_min(res, k * d / (t + c->shadowBias));
    t += cl_max(d, c->eps);
    if (t >= lmax) break;
  }
  return cl_clamp(res, 0.0f, 1.0f);
}

static void scene_distance(ctx_t* c, v3 rpos, v3 dir, int steps, int smooth, float* dist,
                           float* code, v3* nrm) {
  c->st.dts_calls++;
  if (c->sdf) { scene_distance_sdf(c, rpos, dir, dist, code, nrm); return; }
  const float h = rpos.y + c->groundY;
  float rd, rc;
  if (h < 1e5f) { rd = h; rc = h; } else { rd = 1e5f; rc = -1.0f; }
  *nrm = ((double)rd < 1e5) ? V(0.0f, 1.0f, 0.0f) : neg(dir);
  const float t_in = box_entry(c, rpos, dir);
  if (t_in >= 0.0f && t_in < rd) {
    const float sf = (float)steps * 0.5f;
    const v3 delta = mul(V(dir.x / sf, dir.y / sf, dir.z / sf), c->ivs);
    v3 p = add(rpos, c->vb);
    if (t_in > 0.0f) p = mad3s(dir, t_in, p);
    p = mul(p, c->ivs);
    while (--steps >= 0) {
      const int v = voxel_at(c, p);
      if (v < 0) break;
      if (v > c->isoVal) {
        int qx, qy, qz;
        cell_of(c, p, &qx, &qy, &qz);
        *nrm = smooth ? smooth_gradient(c, qx, qy, qz) : normalize3(cell_gradient(c, qx, qy, qz));
        const v3 hit = mad3(p, c->vb2, neg(c->vb));
        const float d = length3(sub(rpos, hit)) - c->voxelSize;
        if (d < rd) { rd = d; rc = band(v); }
        *dist = rd; *code = rc;
        return;
      }
      p = add(p, delta);
    }
  }
  *dist = rd; *code = rc;
}

/* ---- outer march: renderer.cl:239-257 ---- */
typedef struct { v3 pos, normal; float distance; int objectID; } hit_t;
static void march(ctx_t* c, v3 ro, v3 rd, hit_t* r, float maxDist, int maxSteps, int smooth) {
  c->st.rays++;
  r->distance = c->startDist;
  while (--maxSteps >= 0) {
    r->pos = mad3s(rd, r->distance, ro); /* ro + rd*distance: same value, the sum commutes */
    float sd, sc;
    scene_distance(c, r->pos, rd, c->maxVoxelIter, smooth, &sd, &sc, &r->normal);
    r->objectID = cl_f2i(sc);
    if (cl_fabs(sd) <= c->eps || r->distance >= maxDist) break;
    r->distance += sd;
  }
  if (r->distance >= maxDist) {
    r->pos = mad3s(rd, r->distance, ro);
    r->objectID = -1;
    r->distance = 1000.0f;
  }
}

/* renderer.cl:259-261 */
static v3 sky(const ctx_t* c, v3 dir) { return mix3s(c->sky1, c->sky2, dir.y * 0.5f + 0.5f); }

/* per-sample state: renderer.cl:27-33 */
typedef struct { v3 eye; v4 mcPos; v3 mcNormal; float px, py; } sample_t;

/* The (uint) casts of the seed expressions (renderer.cl:267, 334, 471, 472) are undefined for
 * values outside [0, 2^32).  Default: the x86-64 lowering (cl_f2u: wraps, what an OpenCL CPU
 * device does).  rmo_set_seed_cast(1): the lowering of GPU devices (gfx950 v_cvt_u32_f32; NVIDIA
 * cvt.rzi.u32.f32): saturate to [0, 2^32 - 1], NaN -> 0 -- used to compare with the reference
 * kernel compiled for gfx950 (oracle/_ref/renderer_gfx950_*.hsaco). */
static int g_seed_cast_gpu = 0;
void rmo_set_seed_cast(int gpu) { g_seed_cast_gpu = gpu; }
int rmo_get_seed_cast(void) { return g_seed_cast_gpu; }
static inline uint32_t seed_cast(float x) {
  if (!g_seed_cast_gpu) return cl_f2u(x);
  if (!(x > 0.0f)) return 0u;            /* negatives, -0, NaN */
  if (x >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)x;
}

/* jittered light position: renderer.cl:263-269 (one seed for all lights of a sample) */
static v3 light_at(ctx_t* c, const sample_t* s, int i) {
  const uint32_t seed = seed_cast(s->px * 1957.0f + s->py * 2173.0f + c->time * 4763.742f);
  const v4 r = table(c, seed);
  return mad3s(V(r.x, r.y, r.z), c->lightScatter, light_pos_opt(c, i));
}

/* renderer.cl:271-273 */
static v3 reflect3(v3 v, v3 n) { const float k = 2.0f * dot3(v, n); return sub(v, muls(n, k)); }

/* fog towards the sky colour + light flares: renderer.cl:275-290 */
static v3 atmosphere(ctx_t* c, const sample_t* s, v3 ro, v3 rd, float dist, v3 col) {
  const float fa = 1.0f - cl_exp(dist * dist * -c->fogPow);
  const v3 sk = sky(c, rd);
  col = V((sk.x - col.x) * fa + col.x, (sk.y - col.y) * fa + col.y, (sk.z - col.z) * fa + col.z);
  for (int i = 0; i < c->numLights; i++) {
    v3 lp = light_at(c, s, i);
    const float d = cl_clamp(dot3(sub(lp, ro), rd), 0.0f, dist);
    lp = mad3s(rd, d, sub(ro, lp));
    const float k = c->flareAmp / dot3(lp, lp);
    col = mad3s(light_color_opt(c, i), k, col);
  }
  return col;
}

/* renderer.cl:292-301 */
static float shadow_term(ctx_t* c, v3 p, v3 ldir, float lmax) {
  hit_t h;
  memset(&h, 0, sizeof h);
  march(c, p, ldir, &h, lmax, c->shadowIter, 0);
  return cl_step(lmax, h.distance);
}
/* renderer.cl:304-311 */
static float schlick(float r0, float smooth, v3 n, v3 view) {
  const float d = cl_clamp(1.0f - dot3(n, neg(view)), 0.0f, 1.0f);
  if (d > 0.0f) {
    const float d2 = d * d;
    return (1.0f - r0) * (smooth * d2 * d2 * d) + r0;
  }
  return 0.0f;
}
/* renderer.cl:313-315 */
static float diffuse_term(v3 ldir, v3 n) { return cl_max(0.0f, dot3(ldir, n)); }
/* renderer.cl:317-325 */
static float blinn_phong(float smooth, v3 raydir, v3 ldir, v3 n) {
  const float nh = dot3(normalize3(sub(ldir, raydir)), n);
  if (nh > 0.0f) {
    const float sp = cl_exp2(6.0f * smooth + 4.0f);
    return cl_pow(nh, sp) * (sp + 2.0f) * 0.125f;
  }
  return 0.0f;
}

/* renderer.cl:327-346 */
static float occlusion(ctx_t* c, v3 pos, v3 normal) {
  c->st.ao_calls++;
  float ao = 1.0f;
  float d = 0.0f;
  uint32_t seed = seed_cast(pos.x * 3183.75f + pos.y * 1831.42f + pos.z * 2945.87f + c->time * 2671.918f);
  for (int i = 0; i <= c->aoIter && (double)ao > 0.01; i++) {
    d += c->aoStepDist;
    seed += 37u;
    const v4 r = table(c, seed);
    const v3 n = normalize3(mad3s(V(r.x, r.y, r.z), 0.2f, normal));
    float sd, sc; v3 nn;
    scene_distance(c, mad3s(n, d, pos), n, c->maxVoxelIter / 2, 0, &sd, &sc, &nn);
    ao *= 1.0f - cl_max((d - sd) * c->aoAmp / d, 0.0f);
  }
  return ao;
}

/* renderer.cl:348-381 */
static v3 lighting(ctx_t* c, const sample_t* s, v3 raydir, v3 hitpos, const mat_t* m, v3 normal,
                   v3 reflectCol) {
  const float ao = occlusion(c, hitpos, normal);
  v3 diff = muls(sky(c, normal), ao);
  v3 spec = muls(reflectCol, ao);
  v3 out = V(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < c->numLights; i++) {
    const v3 dl = sub(light_at(c, s, i), hitpos);
    const float d2 = dot3(dl, dl);
    const float att = 1.0f / d2;
    if (att > c->minLightAtt) {
      const v3 ldir = normalize3(dl);
      const float lmax = cl_min(cl_sqrt(d2) - c->shadowBias, c->maxDist);
      const float sh = c->sdf ? soft_shadow_sdf(c, mad3s(ldir, c->shadowBias, hitpos), ldir, lmax)
                              : shadow_term(c, mad3s(ldir, c->shadowBias, hitpos), ldir, lmax);
      if (sh > 0.0f) {
        const v3 inc = muls(muls(light_color_opt(c, i), sh), att);
        diff = add(diff, muls(inc, diffuse_term(ldir, normal)));
        spec = add(spec, muls(inc, blinn_phong(m->smoothness, raydir, ldir, normal)));
      }
    }
    diff = mul(diff, m->albedo);
    out = add(out, mix3s(diff, spec, schlick(m->r0, m->smoothness, normal, raydir)));
  }
  const float nl = (float)c->numLights;
  return V(out.x / nl, out.y / nl, out.z / nl);
}

/* one reflection bounce: renderer.cl:383-405 (overwrites *h) */
static v3 bounce_colour(ctx_t* c, const sample_t* s, v3 ro, v3 rd, hit_t* h) {
  march(c, ro, rd, h, c->maxDist, c->maxIter, 0);
  v3 col;
  if (h->objectID < 0) {
    col = sky(c, rd);
  } else {
    const mat_t m = material(c, h->objectID);
    col = lighting(c, s, rd, h->pos, &m, h->normal, sky(c, reflect3(rd, h->normal)));
  }
  return atmosphere(c, s, ro, rd, h->distance, col);
}

/* primary ray shading: renderer.cl:407-446 */
static v3 sample_colour(ctx_t* c, const sample_t* s, v3 ro, v3 rd) {
  hit_t h;
  memset(&h, 0, sizeof h);
  march(c, ro, rd, &h, c->maxDist, c->maxIter, 1);
  v3 col;
  if (h.distance >= c->maxDist) {
    col = sky(c, rd);
  } else {
    c->st.primary_hits++;
    const mat_t m = material(c, h.objectID);
    const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
    const v3 norm = mad3s(s->mcNormal, k, h.normal);
    v3 refl = V(0.0f, 0.0f, 0.0f);
    if (m.r0 > 0.0f && c->reflectIter > 0) {
      hit_t rh;
      memset(&rh, 0, sizeof rh);
      rh.pos = h.pos;
      rh.normal = norm;
      v3 dir = rd;
      for (int i = 0; i < c->reflectIter; i++) {
        dir = reflect3(dir, rh.normal);
        const v3 from = mad3s(dir, 0.0075f, rh.pos);
        refl = add(refl, bounce_colour(c, s, from, dir, &rh));
        if (rh.objectID < 0) break;
        if ((double)material(c, rh.objectID).r0 < 0.001) break;
      }
    } else {
      refl = sky(c, reflect3(rd, norm));
    }
    col = lighting(c, s, rd, h.pos, &m, norm, refl);
  }
  return atmosphere(c, s, ro, rd, h.distance, col);
}

/* renderer.cl:467-476 */
static sample_t sample_init(ctx_t* c, int id) {
  sample_t s;
  const float fx = (float)(id % c->resx), fy = (float)(id / c->resx);
  s.mcPos = table(c, (uint32_t)id * 17u + seed_cast(c->time * 3141.3862f));
  const v4 t = table(c, (uint32_t)id * 37u + seed_cast(c->time * 1859.1467f));
  s.mcNormal = normalize3(V(t.x, t.y, t.z));
  s.px = fx + s.mcPos.z;
  s.py = fy + s.mcPos.w;
  s.eye = mad3s(V(s.mcNormal.z, s.mcNormal.x, s.mcNormal.y), c->dof, c->eyePos);
  return s;
}
/* renderer.cl:456-465 */
static v3 camera_dir(const ctx_t* c, const sample_t* s) {
  const v3 fwd = normalize3(sub(c->targetPos, s->eye));
  const v3 right = normalize3(cross3(fwd, c->up));
  float vx = s->px / (float)c->resx * c->fov - c->fov * 0.5f;
  float vy = s->py / (float)c->resy * c->fov - c->fov * 0.5f;
  vy *= -c->invAspect;
  const v3 upv = cross3(right, fwd);
  return normalize3(add(add(muls(right, vx), muls(upv, vy)), fwd));
}

/* one work-item of RenderImage: renderer.cl:478-494 */
static void render_one(ctx_t* c, float* pixels, int id) {
  const sample_t s = sample_init(c, id);
  const v3 rd = camera_dir(c, &s);
  const v3 col = muls(sample_colour(c, &s, s.eye, rd), c->exposure);
  float* px = pixels + 4 * (size_t)id;
  px[0] = px[0] + (col.x - px[0]) * c->frameBlend;
  px[1] = px[1] + (col.y - px[1]) * c->frameBlend;
  px[2] = px[2] + (col.z - px[2]) * c->frameBlend;
  px[3] = 1.0f;
}

static void stats_add(rmo_stats* a, const rmo_stats* b) {
  a->vox_reads += b->vox_reads; a->mc_reads += b->mc_reads; a->rays += b->rays;
  a->dts_calls += b->dts_calls; a->march_steps += b->march_steps; a->ao_calls += b->ao_calls;
  a->primary_hits += b->primary_hits; a->oob_material += b->oob_material;
}

/* ------------------------------------------------------------------ API */

int rmo_hw_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* RenderImage for work-items id0 <= id < id1 (id < n as in the kernel guard).
 * threads <= 0: all hardware threads.  stats (nullable) is ADDED to. */
static void render_image_impl(const uint8_t* vox, const float* mc, const void* opts544,
                              float* pixels, int n, int id0, int id1, int threads,
                              rmo_stats* stats, uint8_t* undefined_mask, const float* sdf) {
  if (id1 > n) id1 = n;
  if (id0 < 0) id0 = 0;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
  rmo_stats total;
  memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(threads)
  {
    ctx_t c;
    ctx_init(&c, vox, mc, opts544);
    c.sdf = sdf;
#pragma omp for schedule(dynamic, 64)
    for (int id = id0; id < id1; id++) {
      const uint64_t before = c.st.oob_material;
      render_one(&c, pixels, id);
      if (undefined_mask && c.st.oob_material != before) undefined_mask[id] = 1;
    }
#pragma omp critical
    stats_add(&total, &c.st);
  }
  if (stats) stats_add(stats, &total);
}

void rmo_render_image(const uint8_t* vox, const float* mc, const void* opts544, float* pixels,
                      int n, int id0, int id1, int threads, rmo_stats* stats) {
  render_image_impl(vox, mc, opts544, pixels, n, id0, id1, threads, stats, NULL, NULL);
}
/* Same; additionally sets undefined_mask[id] = 1 for every work-item that
 * indexed materials[] outside the option record -- behaviour the reference
 * leaves undefined (it reads whatever follows its private copy on the stack),
 * so such samples are excluded from parity statements. */
void rmo_render_image_masked(const uint8_t* vox, const float* mc, const void* opts544,
                             float* pixels, int n, int id0, int id1, int threads,
                             rmo_stats* stats, uint8_t* undefined_mask) {
  render_image_impl(vox, mc, opts544, pixels, n, id0, id1, threads, stats, undefined_mask, NULL);
}

/* TonemapImage: renderer.cl:448-454, 496-508 */
void rmo_tonemap_image(const float* pixels, const void* opts544, uint32_t* argb, int n, int id0,
                       int id1) {
  const float g = ldf((const uint8_t*)opts544, O_gamma);
  if (id1 > n) id1 = n;
  for (int id = id0 < 0 ? 0 : id0; id < id1; id++) {
    uint32_t ch[3];
    for (int k = 0; k < 3; k++) {
      const float x = pixels[4 * (size_t)id + k];
      const float t = x / (g + x);
      const float v = t * t * 255.0f;
      ch[k] = (uint32_t)cl_f2i(cl_clamp(v, 0.0f, 255.0f));
    }
    argb[id] = 0xff000000u | (ch[0] << 16) | (ch[1] << 8) | ch[2];
  }
}

/* The whole pipeline of core.clj:76-97: accumulator zeroed, `iter` passes in
 * order with (opts_i, mc_i), tonemap with opts_0.  pixels: n*4 floats (out),
 * argb: n uint32 (nullable). */
void rmo_render_frame(const uint8_t* vox, const void* opts_array, const float* mc_array, int iter,
                      float* pixels, uint32_t* argb, int n, int threads, rmo_stats* stats) {
  memset(pixels, 0, sizeof(float) * 4 * (size_t)n);
  for (int i = 0; i < iter; i++)
    rmo_render_image(vox, mc_array + (size_t)i * 0x4000 * 4,
                     (const uint8_t*)opts_array + (size_t)i * OPTS_SIZE, pixels, n, 0, n, threads,
                     stats);
  if (argb) rmo_tonemap_image(pixels, opts_array, argb, n, 0, n);
}

/* The pipeline restricted to a list of work-items (sampled parity checks at sizes whose
 * full frame the CPU cannot render in seconds): for every id in ids[0..count) the `iter`
 * passes are applied in order to pixels[4*id..] (zeroed first); other pixels are untouched.
 * undefined_mask (nullable, n bytes) as in rmo_render_image_masked. */
void rmo_render_frame_ids(const uint8_t* vox, const void* opts_array, const float* mc_array, int iter,
                          float* pixels, int n, const int32_t* ids, int count, int threads,
                          uint8_t* undefined_mask) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel num_threads(threads)
  {
    for (int i = 0; i < iter; i++) {
      ctx_t c;
      ctx_init(&c, vox, mc_array + (size_t)i * 0x4000 * 4, (const uint8_t*)opts_array + (size_t)i * OPTS_SIZE);
#pragma omp for schedule(dynamic, 16)
      for (int k = 0; k < count; k++) {
        const int id = ids[k];
        if (id < 0 || id >= n) continue;
        if (i == 0) memset(pixels + 4 * (size_t)id, 0, 16);
        const uint64_t before = c.st.oob_material;
        render_one(&c, pixels, id);
        if (undefined_mask && c.st.oob_material != before) undefined_mask[id] = 1;
      }
      /* (implicit barrier of the omp for: pass i+1 of an id starts after its pass i) */
    }
  }
}

/* QUALITY MODE frame: the pipeline above over a float distance field (rx*ry*rz floats, x
 * fastest, voxelRes of the records) instead of the byte grid.  Not reference-equivalent. */
void rmo_render_sdf_frame(const float* sdf, const void* opts_array, const float* mc_array, int iter,
                          float* pixels, uint32_t* argb, int n, int threads) {
  memset(pixels, 0, sizeof(float) * 4 * (size_t)n);
  for (int i = 0; i < iter; i++)
    render_image_impl(NULL, mc_array + (size_t)i * 0x4000 * 4,
                      (const uint8_t*)opts_array + (size_t)i * OPTS_SIZE, pixels, n, 0, n, threads, NULL,
                      NULL, sdf);
  if (argb) rmo_tonemap_image(pixels, opts_array, argb, n, 0, n);
}

/* scalar built-ins exported for the device-vs-host primitive tests */
float rmo_exp(float x) { return cl_exp(x); }
float rmo_exp2(float x) { return cl_exp2(x); }
float rmo_pow(float x, float y) { return cl_pow(x, y); }
int32_t rmo_f2i(float x) { return cl_f2i(x); }
uint32_t rmo_f2u(float x) { return cl_f2u(x); }
int32_t rmo_convert_int_sat(float x) { return cl_convert_int_sat(x); }
