// rm_stream.hip -- the render path as a STREAM of ray tasks in HBM queues, traced
// by a persistent, vote-scheduled ray engine.
//
// The reference computes a pixel sample as one deeply nested work-item
// (renderer.cl:407-446: primary march, <= 3 reflection bounces, and for every
// shaded point aoIter+1 ambient-occlusion probes + one shadow march per light).
// On a 64-wide wavefront that nest runs at ~1/3 lane utilisation with ~145
// VGPRs, and a one-ray-per-lane kernel still idles 5 of 6 lanes because march
// lengths differ wildly (profiles/r01_stream_v1_pmc.txt).
//
// All rays of a sample are pure functions of data known before they are traced
// (an AO probe needs the shaded point and normal, a shadow ray the point and the
// light, bounce k+1 the hit of bounce k), so the sample is unrolled into
// homogeneous tasks:
//
//   gen_kernel      1 lane / sample : camera ray -> primary queue        (dense)
//   engine<PRIMARY> ray engine over the primary queue -> hit records
//   emit_kernel(L)  1 lane / sample : the hit of level L -> its AO probes, its
//                   shadow rays, and the next reflection ray               (dense)
//   engine<BOUNCE>  ray engine over the bounce queue (x reflectIter, with emit_kernel)
//   engine<SHADOW>, engine<PROBE>  over everything the emits queued
//   combine_kernel  1 lane / sample : replays the reference's shading arithmetic
//                   in its original order from the stored results         (dense)
//
// The ray engine is one persistent loop per wavefront.  A lane owns one ray and
// is in one of six states (refill / outer march step / exact slab test / voxel
// walk / surface decode / write result).  Every turn the wave executes ONLY the
// state most of its lanes are in (ballot + popcount vote); lanes in other states
// wait their turn; a lane that finishes takes the next ray of the queue
// (wave-aggregated atomic).  Rare expensive events (the 6-division slab test,
// the surface decode) are thereby batched instead of stalling 63 lanes each time
// one lane needs them.  A task carries everything that is constant for the ray
// (step vector, slab-test filter), computed once by the dense emitters.
//
// Speculation is exact: the reference stops probing once ao <= 0.01
// (renderer.cl:338); here all probes are traced and combine_kernel applies the
// same early exit when it folds them in order.  Every float is produced by the
// same IEEE operation sequence as in rm_shade.hpp, so the paths are bit-identical.
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

enum { K_PRIMARY = 0, K_BOUNCE = 1, K_SHADOW = 2, K_PROBE = 3 };
// device uint32 counters: [q] = tasks appended to queue q, [8 + q] = tasks handed out
enum { Q_PRIMARY = 0, Q_BOUNCE0 = 1, Q_BOUNCE1 = 2, Q_SHADOW = 3, Q_PROBE = 4, Q_COUNT = 5 };

struct StreamArgs {
  const uint8_t* __restrict__ vox;
  const uint8_t* __restrict__ dist8;
  const uint32_t* __restrict__ surf32;
  const float4* __restrict__ mc_all;    // tables of the batch's passes: [passes][0x4000]
  const RmOpts* __restrict__ opts_all;  // records of the batch's passes (uniform except .time)
  float4* __restrict__ staging;         // [passes][count]
  float4* __restrict__ cam;             // [samples] primary direction
  float4* __restrict__ hits;            // [levels][samples][3]: (pos,dist) (nrm,obj) (ray dir,-)
  float* __restrict__ ao;               // [levels][kAoMax][samples]
  float* __restrict__ sh;               // [levels][4][samples]
  // queues, 4 float4 per task:
  //  (org.xyz, maxDist) (dir.xyz, dest bits) (delta.xyz, inv_s) (near0, far0, slack | <0 = no filter, -)
  float4* __restrict__ q[Q_COUNT];
  unsigned int* __restrict__ counters;
  int n, resx, passes, count;  // count = tiles_per_part*64 lanes per pass
  int samples;                 // passes * count
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
// detail: 0 = seeds only, 1 = + mcNormal/eye, 2 = + camera direction
__device__ __forceinline__ SampleCtx sample_ctx(const StreamArgs& a, int id, int pass, int detail) {
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
  if (detail >= 1) {
    const float4 tn = tab(a, pass, (uint32_t)id * 37u + rmd::f2u(t * 1859.1467f));
    c.mcNormal = rmk::normalize(V(tn.x, tn.y, tn.z));
    c.eye = rmk::mads(V(c.mcNormal.z, c.mcNormal.x, c.mcNormal.y), o.dof, rmk::ld3(o.eyePos));
  }
  if (detail >= 2) {
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

// ---- task construction: everything that is constant along the ray.
// Slab-test filter: the entry / exit parameters of ro + t*rd against the clip box
// are (near0 - t, far0 - t); an approximate copy decides the reference's test
// (renderer.cl:153-161, :214) whenever the outcome is farther than `slack` from
// flipping.  slack < 0 disables the filter for the ray (tiny direction component).
__device__ __forceinline__ void write_task(const StreamArgs& a, int q, unsigned int at, v3 org, v3 dir,
                                           float maxDist, int dest, int walk_steps) {
  const RmOpts& o = a.opts_all[0];
  const float sf = (float)walk_steps * 0.5f;  // renderer.cl:215
  const v3 delta = V(dir.x / sf, dir.y / sf, dir.z / sf) * rmk::ld3(o.invVoxelScale);
  const float s = fmaxf(fmaxf(__builtin_fabsf(delta.x) * (float)o.voxelRes[0],
                              __builtin_fabsf(delta.y) * (float)o.voxelRes[1]),
                        __builtin_fabsf(delta.z) * (float)o.voxelRes[2]);
  const float inv_s = 0.98f * __builtin_amdgcn_rcpf(fmaxf(s, 1e-6f));
  const float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
  const bool ok = fminf(fminf(ax, ay), az) >= 1e-3f && fmaxf(fmaxf(ax, ay), az) <= 2.0f &&
                  fmaxf(fmaxf(__builtin_fabsf(org.x), __builtin_fabsf(org.y)), __builtin_fabsf(org.z)) <= 64.0f;
  const float ix = __builtin_amdgcn_rcpf(dir.x), iy = __builtin_amdgcn_rcpf(dir.y),
              iz = __builtin_amdgcn_rcpf(dir.z);
  const float lx = (o.voxelBoundsMin[0] - org.x) * ix, hx = (o.voxelBoundsMax[0] - org.x) * ix;
  const float ly = (o.voxelBoundsMin[1] - org.y) * iy, hy = (o.voxelBoundsMax[1] - org.y) * iy;
  const float lz = (o.voxelBoundsMin[2] - org.z) * iz, hz = (o.voxelBoundsMax[2] - org.z) * iz;
  const float near0 = fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fminf(lz, hz));
  const float far0 = fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fmaxf(lz, hz));
  // positions are rounded to ~4e-6 and divided by >= 1e-3, quotients (< 7e4) to ~8e-3
  const float slack = ok ? 0.03f + 8e-6f * (__builtin_fabsf(near0) + __builtin_fabsf(far0)) : -1.0f;
  float4* t = a.q[q] + (size_t)at * 4;
  t[0] = make_float4(org.x, org.y, org.z, maxDist);
  t[1] = make_float4(dir.x, dir.y, dir.z, __int_as_float(dest));
  t[2] = make_float4(delta.x, delta.y, delta.z, inv_s);
  t[3] = make_float4(near0, far0, slack, 0.0f);
}

__device__ __forceinline__ float4* hit_rec(const StreamArgs& a, int level, int s) {
  return a.hits + ((size_t)level * a.samples + s) * 3;
}

// ---- stage: camera rays (renderer.cl:467-476, :456-465)
__global__ __launch_bounds__(256) void gen_kernel(StreamArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pass = blockIdx.y;
  if (idx >= a.count) return;
  const int id = pixel_of(a, idx);
  if (id < 0) return;
  const int s = pass * a.count + idx;
  const RmOpts& o = a.opts_all[0];
  const SampleCtx c = sample_ctx(a, id, pass, 2);
  a.cam[s] = make_float4(c.rd0.x, c.rd0.y, c.rd0.z, 0.0f);
  const unsigned int at = push_slots(a.counters + Q_PRIMARY, 1);
  write_task(a, Q_PRIMARY, at, c.eye, c.rd0, o.maxDist, s, o.maxVoxelIter);
}

// ---- stage: tasks of the point hit at `level` (0 = primary hit, k = bounce k)
// renderer.cl:327-346 probes, :361-369 shadow rays, :426-437 next reflection ray
__global__ __launch_bounds__(256) void emit_kernel(StreamArgs a, int level, int bq_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pass = blockIdx.y;
  if (idx >= a.count) return;
  const int id = pixel_of(a, idx);
  if (id < 0) return;
  const int s = pass * a.count + idx;
  const RmOpts& o = a.opts_all[0];
  const float4* h = hit_rec(a, level, s);
  const float4 ha = h[0], hb = h[1];
  const int obj = __float_as_int(hb.w);
  const v3 pos = V(ha.x, ha.y, ha.z);
  v3 nrm = V(hb.x, hb.y, hb.z), dir;
  bool bounce;
  const SampleCtx c = sample_ctx(a, id, pass, level == 0 ? 1 : 0);
  if (level == 0) {
    if (ha.w >= o.maxDist) return;  // primary miss (renderer.cl:415)
    const rmk::Material m = rmk::material_of(o, obj);
    const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
    nrm = rmk::mads(c.mcNormal, k, nrm);  // renderer.cl:420
    const float4 cd = a.cam[s];
    dir = V(cd.x, cd.y, cd.z);
    bounce = m.r0 > 0.0f && o.reflectIter > 0;
  } else {
    if (obj < 0) return;  // no such bounce, or it left the scene (renderer.cl:436)
    const float4 hd = h[2];
    dir = V(hd.x, hd.y, hd.z);
    bounce = level < o.reflectIter && level < a.levels - 1 &&
             !((double)rmk::material_of(o, obj).r0 < 0.001);  // renderer.cl:432, :437
  }
  // AO probes: all aoIter+1 of them (the early exit is applied when they are folded)
  const int nprobe = min(o.aoIter + 1, kAoMax);
  if (nprobe > 0) {
    const unsigned int base = push_slots(a.counters + Q_PROBE, nprobe);
    uint32_t seed = rmd::f2u(pos.x * 3183.75f + pos.y * 1831.42f + pos.z * 2945.87f + c.time * 2671.918f);
    float d = 0.0f;
    for (int i = 0; i < nprobe; i++) {
      d += o.aoStepDist;
      seed += 37u;
      const float4 r = tab(a, pass, seed);
      const v3 nn = rmk::normalize(rmk::mads(V(r.x, r.y, r.z), 0.2f, nrm));
      write_task(a, Q_PROBE, base + i, rmk::mads(nn, d, pos), nn, 0.0f,
                 (level * kAoMax + i) * a.samples + s, o.maxVoxelIter / 2);
    }
  }
  const int nl = o.numLights;
  for (int i = 0; i < nl; i++) {
    const v3 dl = light_at(a, pass, c.lseed, i) - pos;
    const float d2 = rmk::dot(dl, dl);
    const float att = 1.0f / d2;
    if (att > o.minLightAtt) {
      const v3 ldir = rmk::normalize(dl);
      const unsigned int at = push_slots(a.counters + Q_SHADOW, 1);
      write_task(a, Q_SHADOW, at, rmk::mads(ldir, o.shadowBias, pos), ldir,
                 rmd::fmin_cl(rmd::sqrt_rn(d2) - o.shadowBias, o.maxDist), (level * 4 + i) * a.samples + s,
                 o.maxVoxelIter);
    }
  }
  if (bounce) {
    const v3 d = rmk::reflect_of(dir, nrm);     // renderer.cl:433
    const v3 org = rmk::mads(d, 0.0075f, pos);  // renderer.cl:434
    const unsigned int at = push_slots(a.counters + bq_out, 1);
    write_task(a, bq_out, at, org, d, o.maxDist, s, o.maxVoxelIter);
  }
}

// =========================== the ray engine ===========================
enum : int { E_IDLE = 0, E_OUTER, E_NEEDX, E_WALK, E_HIT };

template <int KIND>
__global__ __launch_bounds__(256) void engine_kernel(StreamArgs a, int q, int level) {
  const RmOpts& o = a.opts_all[0];
  const unsigned int total = a.counters[q];
  unsigned int* __restrict__ head = a.counters + 8 + q;
  const float4* __restrict__ tasks = a.q[q];
  const int lane = threadIdx.x & 63;
  const int walk_steps = KIND == K_PROBE ? o.maxVoxelIter / 2 : o.maxVoxelIter;
  const int outer_steps = KIND == K_PROBE ? 1 : (KIND == K_SHADOW ? o.shadowIter : o.maxIter);
  const bool smooth = KIND == K_PRIMARY;

  int st = E_IDLE;
  bool drained = false;  // wave-uniform: the queue has no more tasks
  v3 ro = V(0, 0, 0), rd = V(0, 0, 0), delta = V(0, 0, 0), nrm = V(0, 0, 0), p = V(0, 0, 0);
  float maxDist = 0, inv_s = 0, near0 = 0, far0 = 0, slack = -1, dist = 0, last_t = 0;
  float est_sd = 0, est_code = 0, g_rd = 0, g_rc = 0;
  int dest = 0, osteps = 0, obj = 0, wsteps = 0, cell = 0;
  bool have_est = false;

  for (;;) {
    // ---- schedule.  The two common states (outer march step, voxel walk) run every
    // turn for whoever is in them; the rare expensive ones (queue refill, exact slab
    // test, surface decode) run only once enough lanes wait for them -- or nothing
    // else is left to do -- so one lane's event never stalls the other 63.
    const int nI = __popcll(__ballot(st == E_IDLE));
    const int nO = __popcll(__ballot(st == E_OUTER));
    const int nX = __popcll(__ballot(st == E_NEEDX));
    const int nW = __popcll(__ballot(st == E_WALK));
    const int nH = __popcll(__ballot(st == E_HIT));
    const int busy = nO + nW;
    if (busy + nX + nH == 0 && (drained || nI == 0)) break;

    if (!drained && (nI >= 16 || (nI > 0 && busy + nX + nH == 0))) {  // ---- refill
      const unsigned long long need = __ballot(st == E_IDLE);
      const int cnt = __popcll(need);
      unsigned int base = 0;
      const int leader = __ffsll((long long)need) - 1;
      if (lane == leader) base = atomicAdd(head, (unsigned int)cnt);
      base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
      if (base + (unsigned int)cnt >= total) drained = true;
      if (st == E_IDLE) {
        const unsigned int i = base + (unsigned int)__popcll(need & ((1ull << lane) - 1ull));
        if (i < total) {
          const float4* t = tasks + (size_t)i * 4;
          const float4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
          ro = V(t0.x, t0.y, t0.z); maxDist = t0.w;
          rd = V(t1.x, t1.y, t1.z); dest = __float_as_int(t1.w);
          delta = V(t2.x, t2.y, t2.z); inv_s = t2.w;
          near0 = t3.x; far0 = t3.y; slack = t3.z;
          dist = o.startDist;  // renderer.cl:242
          last_t = dist;
          osteps = outer_steps;
          have_est = false;
          obj = 0;
          st = E_OUTER;
        }
      }
    }

    if (nO > 0 && st == E_OUTER) {  // ---- outer march: a few steps while they stay cheap
#pragma unroll 1
      for (int turn = 0; turn < 3 && st == E_OUTER; turn++) {
        bool fin = false;
        if (have_est) {  // renderer.cl:246-250
          have_est = false;
          obj = rmd::f2i(est_code);
          if (KIND == K_PROBE) fin = true;
          else if (__builtin_fabsf(est_sd) <= o.eps || dist >= maxDist) fin = true;
          else dist += est_sd;
        }
        if (!fin && --osteps < 0) fin = true;  // renderer.cl:243
        if (fin) {  // renderer.cl:252-256 and the consumer of the ray
          if (KIND == K_PROBE) {
            a.ao[dest] = est_sd;
          } else {
            float t_pos = last_t;
            if (dist >= maxDist) { t_pos = dist; obj = -1; dist = 1000.0f; }
            if (KIND == K_SHADOW) {
              a.sh[dest] = rmd::step_cl(maxDist, dist);  // renderer.cl:300
            } else {
              const v3 pos = rmk::mads(rd, t_pos, ro);
              float4* h = hit_rec(a, level, dest);
              h[0] = make_float4(pos.x, pos.y, pos.z, dist);
              h[1] = make_float4(nrm.x, nrm.y, nrm.z, __int_as_float(obj));
              h[2] = make_float4(rd.x, rd.y, rd.z, 0.0f);
            }
          }
          st = E_IDLE;
        } else {
          last_t = dist;
          const float py = rd.y * dist + ro.y;  // y of renderer.cl:244
          const float h = py + o.groundY;       // renderer.cl:211
          g_rd = h < 1e5f ? h : 1e5f;
          g_rc = h < 1e5f ? h : -1.0f;
          nrm = (g_rd < 1e5f) ? V(0.f, 1.f, 0.f) : -rd;  // renderer.cl:212
          const float m = slack + 8e-6f * __builtin_fabsf(dist);
          const float tn = near0 - dist, tf = far0 - dist;
          if (slack < 0.0f || walk_steps <= 0) {
            st = E_NEEDX;
          } else if (g_rd <= 0.0f || tf < -m || far0 - near0 < -m || tn > g_rd + m) {
            have_est = true;  // renderer.cl:214 is certainly false: ground / sky term
            est_sd = g_rd;
            est_code = g_rc;
          } else if (tn < -m && tf > m && g_rd > m) {
            // certainly inside the clip box: the slab test returns exactly +0
            p = (rmk::mads(rd, dist, ro) + rmk::ld3(o.voxelBounds)) * rmk::ld3(o.invVoxelScale);
            wsteps = walk_steps;
            st = E_WALK;
          } else {
            st = E_NEEDX;
          }
        }
      }
    }

    if ((nX >= 8 || (nX > 0 && busy == 0)) && st == E_NEEDX) {  // ---- renderer.cl:213-218, exact
      const v3 rpos = rmk::mads(rd, dist, ro);
      const float t_in = rmk::box_entry_of(o, rpos, rd);
      if (t_in >= 0.0f && t_in < g_rd && walk_steps > 0) {
        v3 pp = rpos + rmk::ld3(o.voxelBounds);
        if (t_in > 0.0f) pp = rmk::mads(rd, t_in, pp);
        p = pp * rmk::ld3(o.invVoxelScale);
        wsteps = walk_steps;
        st = E_WALK;
      } else {
        have_est = true;
        est_sd = g_rd;
        est_code = g_rc;
        st = E_OUTER;
      }
    }

    if (nW > 0 && st == E_WALK) {  // ---- renderer.cl:219-234, a few lookups per turn
#pragma unroll 1
      for (int turn = 0; turn < 4 && st == E_WALK; turn++) {
        const int r = rmk::walk_step(o, a.dist8, p, wsteps, delta, inv_s, &cell);
        if (r == 1) {
          st = E_HIT;
        } else if (r == 2) {
          have_est = true; est_sd = g_rd; est_code = g_rc; st = E_OUTER;
        }
      }
    }

    if ((nH >= 8 || (nH > 0 && busy == 0)) && st == E_HIT) {  // ---- renderer.cl:222-231 via surf32
      const uint32_t w = a.surf32[cell];
      const int v = (int)(w & 0xffu);
      nrm = rmk::surf_normal(w, smooth);
      const v3 rpos = rmk::mads(rd, last_t, ro);
      const v3 hit = rmk::madv(p, rmk::ld3(o.voxelBounds2), -rmk::ld3(o.voxelBounds));
      const float d = rmk::length(rpos - hit) - o.voxelSize;
      if (d < g_rd) { g_rd = d; g_rc = rmk::band_of(v); }
      have_est = true; est_sd = g_rd; est_code = g_rc;
      st = E_OUTER;
    }
  }
}

// ---- AO probes (renderer.cl:342): one short walk each, uniform enough that a plain
// one-lane-per-probe kernel beats the engine's scheduling overhead (measured)
__global__ __launch_bounds__(256) void probe_kernel(StreamArgs a) {
  const unsigned int total = a.counters[Q_PROBE];
  const RmOpts& o = a.opts_all[0];
  rmk::Scene sc{a.vox, a.mc_all, a.opts_all, a.dist8, a.surf32};
  rmk::Tracer<false, true> tr(sc);
  const float4* __restrict__ tasks = a.q[Q_PROBE];
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float4 t0 = tasks[(size_t)i * 4], t1 = tasks[(size_t)i * 4 + 1];
    float sd, code;
    v3 nn;
    tr.scene_distance(V(t0.x, t0.y, t0.z), V(t1.x, t1.y, t1.z), o.maxVoxelIter / 2, false, sd, code, nn);
    a.ao[__float_as_int(t1.w)] = sd;
  }
}

// ---- final stage: shading arithmetic of the sample, in the reference's order
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
  const SampleCtx c = sample_ctx(a, id, pass, 2);
  const float4* h0 = hit_rec(a, 0, s);
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
        const float4* hk = hit_rec(a, 1 + i, s);
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

inline size_t align_up(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

namespace rmk {

size_t stream_workspace_bytes(int samples, int levels, int num_lights) {
  const size_t S = (size_t)samples;
  size_t b = 4096;
  b += align_up(S * 16);                                // cam
  b += align_up(S * levels * 3 * 16);                   // hits
  b += align_up(S * levels * kAoMax * 4);               // ao
  b += align_up(S * levels * 4 * 4);                    // sh
  b += 3 * align_up(S * 64);                            // primary + 2 bounce queues
  b += align_up(S * levels * (size_t)num_lights * 64);  // shadow queue
  b += align_up(S * levels * (size_t)kAoMax * 64);      // probe queue
  return b;
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
  auto take = [&](size_t bytes) { char* p = w; w += align_up(bytes); return p; };
  a.counters = reinterpret_cast<unsigned int*>(take(4096));
  a.cam = reinterpret_cast<float4*>(take(S * 16));
  a.hits = reinterpret_cast<float4*>(take(S * L.levels * 3 * 16));
  a.ao = reinterpret_cast<float*>(take(S * L.levels * kAoMax * 4));
  a.sh = reinterpret_cast<float*>(take(S * L.levels * 4 * 4));
  a.q[Q_PRIMARY] = reinterpret_cast<float4*>(take(S * 64));
  a.q[Q_BOUNCE0] = reinterpret_cast<float4*>(take(S * 64));
  a.q[Q_BOUNCE1] = reinterpret_cast<float4*>(take(S * 64));
  a.q[Q_SHADOW] = reinterpret_cast<float4*>(take(S * L.levels * (size_t)L.num_lights * 64));
  a.q[Q_PROBE] = reinterpret_cast<float4*>(take(S * L.levels * (size_t)kAoMax * 64));

  hipError_t e = hipMemsetAsync(a.counters, 0, 4096, st);
  if (e != hipSuccess) return e;
  if (L.levels > 1) {  // bounce hit records start out as "no such bounce" (obj = -1)
    e = hipMemsetAsync(a.hits + S * 3, 0xff, S * (L.levels - 1) * 3 * 16, st);
    if (e != hipSuccess) return e;
  }
  const dim3 per_sample((unsigned)((L.count + 255) / 256), (unsigned)L.passes);
  const int grid = L.queue_blocks;
  gen_kernel<<<per_sample, 256, 0, st>>>(a);
  engine_kernel<K_PRIMARY><<<grid, 256, 0, st>>>(a, Q_PRIMARY, 0);
  for (int lvl = 0; lvl < L.levels; lvl++) {
    const int bq_out = (lvl & 1) ? Q_BOUNCE1 : Q_BOUNCE0;
    if (lvl >= 2) {  // this queue was filled two levels ago and is drained: rewind it
      e = hipMemsetAsync(a.counters + bq_out, 0, 4, st);
      if (e == hipSuccess) e = hipMemsetAsync(a.counters + 8 + bq_out, 0, 4, st);
      if (e != hipSuccess) return e;
    }
    emit_kernel<<<per_sample, 256, 0, st>>>(a, lvl, bq_out);
    if (lvl + 1 < L.levels) engine_kernel<K_BOUNCE><<<grid, 256, 0, st>>>(a, bq_out, lvl + 1);
  }
  engine_kernel<K_SHADOW><<<grid, 256, 0, st>>>(a, Q_SHADOW, 0);
  probe_kernel<<<grid, 256, 0, st>>>(a);
  combine_kernel<<<per_sample, 256, 0, st>>>(a);
  return hipGetLastError();
}

}  // namespace rmk
