// rm_stream.hip -- the render path as a STREAM of small dense kernels.
//
// The reference computes a pixel sample as one deeply nested work-item
// (renderer.cl:407-446: primary march, <= 3 reflection bounces, and for every
// shaded point 6 ambient-occlusion probes + one shadow march per light).  On a
// 64-wide wavefront that nest runs at ~1/3 lane utilisation and ~145 VGPRs
// (3 waves per SIMD), and once the fixed-step walk is accelerated (rm_accel.hip)
// it is that control structure, not the voxel fetches, that costs the time.
//
// All the rays of a sample are pure functions of data known BEFORE they are
// traced: an AO probe needs only the shaded point and its normal, a shadow ray
// only the point and the light, bounce k+1 only the hit of bounce k.  The sample
// is therefore unrolled into a small dependency graph whose nodes are homogeneous
// tasks kept in HBM queues, and every kernel below runs ONE kind of task on
// densely packed lanes:
//
//   primary_kernel   1 lane / sample: camera ray + primary march; on a hit it
//                    appends the point's AO probes, its shadow rays and bounce 1
//   bounce_kernel    1 lane / bounce ray (runs reflectIter times, queue ping-pong);
//                    on a hit appends that point's probes / shadow rays / next bounce
//   shadow_kernel    1 lane / shadow ray  -> 0/1 visibility
//   probe_kernel     1 lane / AO probe    -> one distance estimate
//   combine_kernel   1 lane / sample: replays the reference's shading arithmetic
//                    in its original order from the stored results -> staging colour
//
// Speculation is exact: the reference stops probing once ao <= 0.01
// (renderer.cl:338); here all aoIter+1 probes are traced and combine_kernel
// applies the same early exit when it folds them in order.  Every float is
// produced by the same IEEE operation sequence as in rm_shade.hpp, so the two
// paths are bit-identical (tests compare both with the oracle).
//
// Cost of the queues: ~36 B per ray, 32 B per probe, 32 B per stored hit --
// HBM streaming traffic of ~1 KB per hit sample, written and read once.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "rm_kernels.h"
#include "rm_shade.hpp"
#include "rm_stream.h"

namespace {

using rmk::v3;
using rmk::V;
constexpr int kTile = 8;
constexpr int kAoMax = 8;  // probes stored per shaded point (aoIter + 1 <= kAoMax)

struct TileGeom { int tiles_x, tiles_total; };
__host__ __device__ inline TileGeom tile_geom(int resx, int n) {
  const int rows = (n + resx - 1) / resx;
  TileGeom g;
  g.tiles_x = (resx + kTile - 1) / kTile;
  g.tiles_total = g.tiles_x * ((rows + kTile - 1) / kTile);
  return g;
}

// queue counters (device uint32[8])
enum { C_BQ0 = 0, C_BQ1 = 1, C_SQ = 2, C_PQ = 3 };

struct StreamArgs {
  const uint8_t* __restrict__ vox;
  const uint8_t* __restrict__ dist8;
  const uint32_t* __restrict__ surf32;
  const float4* __restrict__ mc_all;    // tables of the batch's passes: [passes][0x4000]
  const RmOpts* __restrict__ opts_all;  // records of the batch's passes (uniform except .time)
  float4* __restrict__ staging;         // [passes][count]
  // per-sample results
  float4* __restrict__ hits;     // [(levels) * 2 * samples]: level-major, (pos,dist),(nrm,obj)
  float* __restrict__ ao;        // [levels][kAoMax][samples]
  float* __restrict__ sh;        // [levels][4][samples]
  // queues
  float4* __restrict__ bq_a[2];  // bounce rays: (org, maxDist)
  float4* __restrict__ bq_b[2];  //              (dir, sample | bounce<<28 as bits)
  float4* __restrict__ sq_a;     // shadow rays: (org, maxDist)
  float4* __restrict__ sq_b;     //              (dir, dest index into sh as bits)
  float4* __restrict__ pq_a;     // probes: (rpos, dest index into ao as bits)
  float4* __restrict__ pq_b;     //         (dir, -)
  unsigned int* __restrict__ counters;
  int n, resx, passes, count;    // count = tiles_per_part*64 lanes per pass
  int samples;                   // passes * count
  int tile_first, tile_stride, levels;
};

// ---- wave-aggregated queue append: `k` consecutive slots for every active lane
__device__ __forceinline__ unsigned int push_slots(unsigned int* counter, int k) {
  const unsigned long long act = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int rank = __popcll(act & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)act) - 1;
  unsigned int base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned int)(k * __popcll(act)));
  base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
  return base + (unsigned int)(rank * k);
}

__device__ __forceinline__ int pixel_of(const StreamArgs& a, int idx) {  // idx in [0,count) -> work-item id or -1
  const TileGeom g = tile_geom(a.resx, a.n);
  const int slot = idx >> 6, lane = idx & 63;
  const long long tile = a.tile_first + (long long)slot * a.tile_stride;
  if (tile >= g.tiles_total) return -1;
  const int tx = (int)(tile % g.tiles_x), ty = (int)(tile / g.tiles_x);
  const int x = tx * kTile + (lane & 7), y = ty * kTile + (lane >> 3);
  if (x >= a.resx) return -1;
  const long long id = (long long)y * a.resx + x;
  return id < a.n ? (int)id : -1;
}

// per-sample values every stage can rebuild from (id, pass): renderer.cl:467-476, :456-465, :267
struct SampleCtx {
  int pass;
  float time;
  uint32_t lseed;
  v3 mcNormal, eye, rd0;
};
__device__ __forceinline__ float4 tab(const StreamArgs& a, int pass, uint32_t seed) {
  return a.mc_all[(size_t)pass * RM_TABLE_ENTRIES + (seed & (RM_TABLE_ENTRIES - 1))];
}
__device__ __forceinline__ SampleCtx sample_ctx(const StreamArgs& a, int id, int pass, bool with_camera) {
  const RmOpts& o = a.opts_all[0];
  SampleCtx c;
  c.pass = pass;
  const float t = a.opts_all[pass].time;
  c.time = t;
  const int resx = o.resolution[0];
  const float fx = (float)(id % resx), fy = (float)(id / resx);
  const float4 mcPos = tab(a, pass, (uint32_t)id * 17u + rmd::f2u(t * 3141.3862f));
  const float px = fx + mcPos.z, py = fy + mcPos.w;
  c.lseed = rmd::f2u(px * 1957.0f + py * 2173.0f + t * 4763.742f);
  c.mcNormal = V(0.f, 0.f, 0.f);
  c.eye = c.mcNormal;
  c.rd0 = c.mcNormal;
  if (with_camera) {
    const float4 tn = tab(a, pass, (uint32_t)id * 37u + rmd::f2u(t * 1859.1467f));
    c.mcNormal = rmk::normalize(V(tn.x, tn.y, tn.z));
    c.eye = rmk::mads(V(c.mcNormal.z, c.mcNormal.x, c.mcNormal.y), o.dof, rmk::ld3(o.eyePos));
    const v3 fwd = rmk::normalize(rmk::ld3(o.targetPos) - c.eye);
    const v3 right = rmk::normalize(rmk::cross(fwd, rmk::ld3(o.up)));
    float vx = px / (float)o.resolution[0] * o.fov - o.fov * 0.5f;
    float vy = py / (float)o.resolution[1] * o.fov - o.fov * 0.5f;
    vy *= -o.invAspect;
    const v3 upv = rmk::cross(right, fwd);
    c.rd0 = rmk::normalize(right * vx + upv * vy + fwd);
  }
  return c;
}
__device__ __forceinline__ v3 light_at(const StreamArgs& a, int pass, uint32_t lseed, int i) {
  const RmOpts& o = a.opts_all[0];
  const float4 r = tab(a, pass, lseed);
  return rmk::mads(V(r.x, r.y, r.z), o.lightScatter, rmk::ld3(o.lightPos[i]));
}

// Append everything the lighting of one shaded point needs (renderer.cl:327-346
// probes, :361-369 shadow rays).  `level` 0 = primary hit, k = bounce k.
__device__ __forceinline__ void emit_point_tasks(const StreamArgs& a, int s, int level, int pass,
                                                 float time, uint32_t lseed, v3 pos, v3 nrm) {
  const RmOpts& o = a.opts_all[0];
  // AO probes: all aoIter+1 of them (the early exit is applied when they are folded)
  const int nprobe = min(o.aoIter + 1, kAoMax);
  if (nprobe > 0) {
    const unsigned int base = push_slots(a.counters + C_PQ, nprobe);
    uint32_t seed = rmd::f2u(pos.x * 3183.75f + pos.y * 1831.42f + pos.z * 2945.87f + time * 2671.918f);
    float d = 0.0f;
    for (int i = 0; i < nprobe; i++) {
      d += o.aoStepDist;
      seed += 37u;
      const float4 r = tab(a, pass, seed);
      const v3 nn = rmk::normalize(rmk::mads(V(r.x, r.y, r.z), 0.2f, nrm));
      const v3 rp = rmk::mads(nn, d, pos);
      const int dest = (level * kAoMax + i) * a.samples + s;
      a.pq_a[base + i] = make_float4(rp.x, rp.y, rp.z, __int_as_float(dest));
      a.pq_b[base + i] = make_float4(nn.x, nn.y, nn.z, 0.0f);
    }
  }
  const int nl = o.numLights;
  for (int i = 0; i < nl; i++) {
    const v3 dl = light_at(a, pass, lseed, i) - pos;
    const float d2 = rmk::dot(dl, dl);
    const float att = 1.0f / d2;
    if (att > o.minLightAtt) {
      const v3 ldir = rmk::normalize(dl);
      const v3 org = rmk::mads(ldir, o.shadowBias, pos);
      const float lmax = rmd::fmin_cl(rmd::sqrt_rn(d2) - o.shadowBias, o.maxDist);
      const unsigned int at = push_slots(a.counters + C_SQ, 1);
      const int dest = (level * 4 + i) * a.samples + s;
      a.sq_a[at] = make_float4(org.x, org.y, org.z, lmax);
      a.sq_b[at] = make_float4(ldir.x, ldir.y, ldir.z, __int_as_float(dest));
    }
  }
}
__device__ __forceinline__ void emit_bounce(const StreamArgs& a, int q, int s, int bounce, v3 dir,
                                            v3 from_pos, v3 nrm) {
  const v3 d = rmk::reflect_of(dir, nrm);          // renderer.cl:433
  const v3 org = rmk::mads(d, 0.0075f, from_pos);  // renderer.cl:434
  const unsigned int at = push_slots(a.counters + (q ? C_BQ1 : C_BQ0), 1);
  a.bq_a[q][at] = make_float4(org.x, org.y, org.z, a.opts_all[0].maxDist);
  a.bq_b[q][at] = make_float4(d.x, d.y, d.z, __int_as_float(s | (bounce << 28)));
}
__device__ __forceinline__ void store_hit(const StreamArgs& a, int level, int s, v3 pos, float dist,
                                          v3 nrm, int obj) {
  float4* h = a.hits + ((size_t)level * a.samples + s) * 2;
  h[0] = make_float4(pos.x, pos.y, pos.z, dist);
  h[1] = make_float4(nrm.x, nrm.y, nrm.z, __int_as_float(obj));
}

using Tr = rmk::Tracer<false, true>;

#ifdef RM_WORK_STATS
// debug: counters[16 + 8*kind + {0 rays,1 iters,2 filtered,3 walks,4 lookups,5 jumps}]
__device__ __forceinline__ void flush_stats(const StreamArgs& a, int kind, Tr& tr, unsigned int rays) {
  unsigned int v[6] = {rays, tr.ws_iters, tr.ws_filtered, tr.ws_walks, tr.ws_lookups, tr.ws_jumps};
  for (int k = 0; k < 6; k++) atomicAdd(a.counters + 16 + 8 * kind + k, v[k]);
}
#define RM_FLUSH(kind, tr, rays) flush_stats(a, kind, tr, rays)
#define RM_WS_RAY (nrays++)
#else
#define RM_FLUSH(kind, tr, rays) ((void)0)
#define RM_WS_RAY ((void)nrays)
#endif

// ---- stage 1: camera ray + primary march (renderer.cl:489-490, :413)
__global__ __launch_bounds__(256) void primary_kernel(StreamArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pass = blockIdx.y;
  if (idx >= a.count) return;
  const int id = pixel_of(a, idx);
  if (id < 0) return;
  const int s = pass * a.count + idx;
  const RmOpts& o = a.opts_all[0];
  const SampleCtx c = sample_ctx(a, id, pass, true);
  rmk::Scene sc{a.vox, a.mc_all, a.opts_all, a.dist8, a.surf32};
  Tr tr(sc);
  Tr::Hit h{};
  tr.march(c.eye, c.rd0, h, o.maxDist, o.maxIter, true);
  store_hit(a, 0, s, h.pos, h.distance, h.normal, h.objectID);
  RM_FLUSH(0, tr, 1);
  if (h.distance >= o.maxDist) return;  // miss: combine_kernel shades the sky
  const rmk::Material m = rmk::material_of(o, h.objectID);
  const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
  const v3 norm = rmk::mads(c.mcNormal, k, h.normal);
  emit_point_tasks(a, s, 0, pass, c.time, c.lseed, h.pos, norm);
  if (m.r0 > 0.0f && o.reflectIter > 0) emit_bounce(a, 0, s, 0, c.rd0, h.pos, norm);
}

// ---- stage 2 (x reflectIter): one reflection bounce (renderer.cl:389, :436-437)
__global__ __launch_bounds__(256) void bounce_kernel(StreamArgs a, int q) {
  unsigned int nrays = 0;
  const unsigned int total = a.counters[q ? C_BQ1 : C_BQ0];
  const RmOpts& o = a.opts_all[0];
  rmk::Scene sc{a.vox, a.mc_all, a.opts_all, a.dist8, a.surf32};
  Tr tr(sc);
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float4 ta = a.bq_a[q][i], tb = a.bq_b[q][i];
    const int word = __float_as_int(tb.w);
    const int s = word & 0x0fffffff, bounce = (word >> 28) & 7;
    const v3 org = V(ta.x, ta.y, ta.z), dir = V(tb.x, tb.y, tb.z);
    Tr::Hit h{};
    tr.march(org, dir, h, o.maxDist, o.maxIter, false);
    store_hit(a, 1 + bounce, s, h.pos, h.distance, h.normal, h.objectID);
    RM_WS_RAY;
    if (h.objectID < 0) continue;
    const int pass = s / a.count, idx = s - pass * a.count;
    const int id = pixel_of(a, idx);
    const SampleCtx c = sample_ctx(a, id, pass, false);
    emit_point_tasks(a, s, 1 + bounce, pass, c.time, c.lseed, h.pos, h.normal);
    const bool more = bounce + 1 < o.reflectIter && bounce + 1 < a.levels - 1 &&
                      !((double)rmk::material_of(o, h.objectID).r0 < 0.001);
    if (more) emit_bounce(a, q ^ 1, s, bounce + 1, dir, h.pos, h.normal);
  }
  RM_FLUSH(1, tr, nrays);
}

// ---- stage 3: shadow rays (renderer.cl:292-301)
__global__ __launch_bounds__(256) void shadow_kernel(StreamArgs a) {
  unsigned int nrays = 0;
  const unsigned int total = a.counters[C_SQ];
  const RmOpts& o = a.opts_all[0];
  rmk::Scene sc{a.vox, a.mc_all, a.opts_all, a.dist8, a.surf32};
  Tr tr(sc);
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float4 ta = a.sq_a[i], tb = a.sq_b[i];
    Tr::Hit h{};
    tr.march(V(ta.x, ta.y, ta.z), V(tb.x, tb.y, tb.z), h, ta.w, o.shadowIter, false);
    a.sh[__float_as_int(tb.w)] = rmd::step_cl(ta.w, h.distance);
    RM_WS_RAY;
  }
  RM_FLUSH(2, tr, nrays);
}

// ---- stage 4: AO probes (renderer.cl:342)
__global__ __launch_bounds__(256) void probe_kernel(StreamArgs a) {
  unsigned int nrays = 0;
  const unsigned int total = a.counters[C_PQ];
  const RmOpts& o = a.opts_all[0];
  rmk::Scene sc{a.vox, a.mc_all, a.opts_all, a.dist8, a.surf32};
  Tr tr(sc);
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float4 ta = a.pq_a[i], tb = a.pq_b[i];
    float sd, code;
    v3 nn;
    tr.scene_distance(V(ta.x, ta.y, ta.z), V(tb.x, tb.y, tb.z), o.maxVoxelIter / 2, false, sd, code, nn);
    a.ao[__float_as_int(ta.w)] = sd;
    RM_WS_RAY;
  }
  RM_FLUSH(3, tr, nrays);
}

// ---- stage 5: shading arithmetic of the sample, in the reference's order
__device__ __forceinline__ v3 atmosphere(const StreamArgs& a, const SampleCtx& c, v3 ro, v3 rdir,
                                         float dist, v3 col) {  // renderer.cl:275-290
  const RmOpts& o = a.opts_all[0];
  const float fa = 1.0f - rmd::exp_det(dist * dist * -o.fogPow);
  const v3 sk = rmk::sky_of(o, rdir);
  col = V((sk.x - col.x) * fa + col.x, (sk.y - col.y) * fa + col.y, (sk.z - col.z) * fa + col.z);
  const int nl = o.numLights;
  for (int i = 0; i < nl; i++) {
    v3 lp = light_at(a, c.pass, c.lseed, i);
    const float d = rmd::clamp_cl(rmk::dot(lp - ro, rdir), 0.0f, dist);
    lp = rmk::mads(rdir, d, ro - lp);
    const float k = o.flareAmp / rmk::dot(lp, lp);
    col = rmk::mads(rmk::ld3(o.lightColor[i]), k, col);
  }
  return col;
}
__device__ __forceinline__ v3 lighting(const StreamArgs& a, const SampleCtx& c, int s, int level, v3 pos,
                                       v3 nrm, v3 raydir, const rmk::Material& m, v3 reflectCol) {
  const RmOpts& o = a.opts_all[0];
  // renderer.cl:332-345 with the probes' distance estimates read back
  float ao = 1.0f, d = 0.0f;
  for (int i = 0; i <= o.aoIter && (double)ao > 0.01; i++) {
    d += o.aoStepDist;
    const float sd = a.ao[(size_t)(level * kAoMax + i) * a.samples + s];
    ao *= 1.0f - rmd::fmax_cl((d - sd) * o.aoAmp / d, 0.0f);
  }
  v3 diff = rmk::sky_of(o, nrm) * ao;
  v3 spec = reflectCol * ao;
  v3 out = V(0.f, 0.f, 0.f);
  const int nl = o.numLights;
  for (int i = 0; i < nl; i++) {  // renderer.cl:361-379
    const v3 dl = light_at(a, c.pass, c.lseed, i) - pos;
    const float d2 = rmk::dot(dl, dl);
    const float att = 1.0f / d2;
    if (att > o.minLightAtt) {
      const v3 ldir = rmk::normalize(dl);
      const float shf = a.sh[(size_t)(level * 4 + i) * a.samples + s];
      if (shf > 0.0f) {
        const v3 inc = (rmk::ld3(o.lightColor[i]) * shf) * att;
        diff = diff + inc * rmd::fmax_cl(0.0f, rmk::dot(ldir, nrm));
        spec = spec + inc * rmk::blinn_phong_of(m.smoothness, raydir, ldir, nrm);
      }
    }
    diff = diff * m.albedo;
    out = out + rmk::mixs(diff, spec, rmk::schlick_of(m.r0, m.smoothness, nrm, raydir));
  }
  const float fl = (float)nl;
  return V(out.x / fl, out.y / fl, out.z / fl);
}

__global__ __launch_bounds__(256) void combine_kernel(StreamArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pass = blockIdx.y;
  if (idx >= a.count) return;
  const int id = pixel_of(a, idx);
  if (id < 0) return;
  const int s = pass * a.count + idx;
  const RmOpts& o = a.opts_all[0];
  const SampleCtx c = sample_ctx(a, id, pass, true);
  const float4* h0 = a.hits + (size_t)s * 2;
  const float4 ha = h0[0], hb = h0[1];
  const float hdist = ha.w;
  v3 col;
  if (hdist >= o.maxDist) {  // renderer.cl:415-416
    col = rmk::sky_of(o, c.rd0);
  } else {
    const v3 hpos = V(ha.x, ha.y, ha.z);
    const rmk::Material m = rmk::material_of(o, __float_as_int(hb.w));
    const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
    const v3 norm = rmk::mads(c.mcNormal, k, V(hb.x, hb.y, hb.z));
    v3 refl = V(0.f, 0.f, 0.f);
    if (m.r0 > 0.0f && o.reflectIter > 0) {  // renderer.cl:426-438
      v3 lpos = hpos, lnrm = norm, dir = c.rd0;
      for (int i = 0; i < o.reflectIter && i < a.levels - 1; i++) {
        dir = rmk::reflect_of(dir, lnrm);
        const v3 from = rmk::mads(dir, 0.0075f, lpos);
        const float4* hk = a.hits + ((size_t)(1 + i) * a.samples + s) * 2;
        const float4 ka = hk[0], kb = hk[1];
        const int obj = __float_as_int(kb.w);
        v3 bc;
        if (obj < 0) {  // renderer.cl:391-392
          bc = rmk::sky_of(o, dir);
        } else {
          const v3 bpos = V(ka.x, ka.y, ka.z), bn = V(kb.x, kb.y, kb.z);
          bc = lighting(a, c, s, 1 + i, bpos, bn, dir, rmk::material_of(o, obj),
                        rmk::sky_of(o, rmk::reflect_of(dir, bn)));
        }
        refl = refl + atmosphere(a, c, from, dir, ka.w, bc);
        if (obj < 0) break;
        if ((double)rmk::material_of(o, obj).r0 < 0.001) break;
        lpos = V(ka.x, ka.y, ka.z);
        lnrm = V(kb.x, kb.y, kb.z);
      }
    } else {
      refl = rmk::sky_of(o, rmk::reflect_of(c.rd0, norm));
    }
    col = lighting(a, c, s, 0, hpos, norm, c.rd0, m, refl);
  }
  col = atmosphere(a, c, c.eye, c.rd0, hdist, col);
  const float e = o.exposure;
  a.staging[(size_t)pass * a.count + idx] = make_float4(col.x * e, col.y * e, col.z * e, 1.0f);
}

}  // namespace

namespace rmk {

size_t stream_workspace_bytes(int samples, int levels, int num_lights) {
  const size_t S = (size_t)samples;
  size_t b = 0;
  b += S * levels * 2 * 16;                 // hits
  b += S * levels * kAoMax * 4;             // ao
  b += S * levels * 4 * 4;                  // sh
  b += 2 * 2 * S * 16;                      // bounce queues (ping-pong, a+b)
  b += 2 * S * levels * (size_t)num_lights * 16;  // shadow queue
  b += 2 * S * levels * kAoMax * 16;        // probe queue
  return b + (64 << 10);  // + per-array alignment slack
}

hipError_t launch_stream_batch(hipStream_t st, const StreamLaunch& L) {
  StreamArgs a;
  a.vox = L.vox;
  a.dist8 = L.accel.dist;
  a.surf32 = L.accel.surf;
  a.mc_all = reinterpret_cast<const float4*>(L.mc);
  a.opts_all = L.opts;
  a.staging = reinterpret_cast<float4*>(L.staging);
  a.n = L.n;
  a.resx = L.resx;
  a.passes = L.passes;
  a.count = L.count;
  a.samples = L.passes * L.count;
  a.tile_first = L.tile_first;
  a.tile_stride = L.tile_stride;
  a.levels = L.levels;
  const size_t S = (size_t)a.samples;
  char* w = static_cast<char*>(L.workspace);
  auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return p; };
  a.counters = reinterpret_cast<unsigned int*>(take(256));
  a.hits = reinterpret_cast<float4*>(take(S * L.levels * 2 * 16));
  a.ao = reinterpret_cast<float*>(take(S * L.levels * kAoMax * 4));
  a.sh = reinterpret_cast<float*>(take(S * L.levels * 4 * 4));
  for (int q = 0; q < 2; q++) {
    a.bq_a[q] = reinterpret_cast<float4*>(take(S * 16));
    a.bq_b[q] = reinterpret_cast<float4*>(take(S * 16));
  }
  const size_t nsq = S * L.levels * (size_t)L.num_lights, npq = S * L.levels * kAoMax;
  a.sq_a = reinterpret_cast<float4*>(take(nsq * 16));
  a.sq_b = reinterpret_cast<float4*>(take(nsq * 16));
  a.pq_a = reinterpret_cast<float4*>(take(npq * 16));
  a.pq_b = reinterpret_cast<float4*>(take(npq * 16));

  hipError_t e = hipMemsetAsync(a.counters, 0, 256, st);
  if (e != hipSuccess) return e;
  const dim3 per_sample((unsigned)((L.count + 255) / 256), (unsigned)L.passes);
  primary_kernel<<<per_sample, 256, 0, st>>>(a);
  const int grid = L.queue_blocks;
  for (int r = 0; r + 1 < L.levels; r++) {
    const int q = r & 1;
    // the queue this round fills was drained two rounds ago: reset its counter first
    e = hipMemsetAsync(a.counters + ((q ^ 1) ? C_BQ1 : C_BQ0), 0, 4, st);
    if (e != hipSuccess) return e;
    bounce_kernel<<<grid, 256, 0, st>>>(a, q);
  }
  shadow_kernel<<<grid, 256, 0, st>>>(a);
  probe_kernel<<<grid, 256, 0, st>>>(a);
  combine_kernel<<<per_sample, 256, 0, st>>>(a);
#ifdef RM_WORK_STATS
  {
    unsigned int hc[64];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hc, a.counters, sizeof hc, hipMemcpyDeviceToHost);
    const char* names[4] = {"primary", "bounce", "shadow", "probe"};
    fprintf(stderr, "[stream stats] samples=%d queues: bq0=%u bq1=%u sq=%u pq=%u\n", a.samples, hc[0], hc[1], hc[2], hc[3]);
    for (int k = 0; k < 4; k++)
      fprintf(stderr, "  %-8s rays=%u iters=%u filtered=%u walks=%u lookups=%u jumps=%u\n", names[k],
              hc[16 + 8 * k], hc[17 + 8 * k], hc[18 + 8 * k], hc[19 + 8 * k], hc[20 + 8 * k], hc[21 + 8 * k]);
  }
#endif
  return hipGetLastError();
}

}  // namespace rmk
