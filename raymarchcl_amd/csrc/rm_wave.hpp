// rm_wave.hpp -- the wave-scheduled form of the render path (gfx950).
//
// Same arithmetic as rm_shade.hpp (every value is produced by the same IEEE
// operation sequence, so results are bit-identical and tests compare the two),
// different execution shape.  The straight form nests the loops the way the
// reference does (renderer.cl:407-446 -> :348-381 -> :327-346 / :292-301 ->
// :239-257 -> :209-237): on a 64-wide wavefront that leaves ~1/3 of the lanes
// active, because neighbouring pixels sit in different nests (one in an AO
// probe, one in a shadow ray, one done) and every nest is separate code.
//
// Here a lane is a small state machine.  ALL ray kinds of all lanes funnel
// into ONE copy of the fixed-step march (the hot loop); everything between two
// marches -- convergence test of the outer march, AO bookkeeping, light loop,
// reflection bounces, fog -- is a short "continuation" that runs only when
// enough lanes are waiting for one.  A wavefront is persistent: it pulls 8x8
// tiles from an atomic queue and keeps a pool of (pixel, pass) samples of its
// tile; a lane that finishes a sample immediately takes the next one from the
// pool (wave-local ballot + prefix count, no atomics), so the tail of
// expensive pixels is filled with other passes of cheap ones.  Sample colours
// go to the staging buffer [pass][tile][lane]; blend_kernel applies the
// reference's in-order frame blend afterwards.
//
// Per-pass option records may differ only in `time` inside one launch (that is
// what the reference host produces, core.clj:99-106); the host splits the
// pass list into such runs.  `time` and the pass's scatter table are per-lane.
#pragma once
#include "rm_shade.hpp"

namespace rmk {

struct WaveArgs {
  const uint8_t* __restrict__ vox;
  const uint8_t* __restrict__ dist8;
  const uint32_t* __restrict__ surf32;
  const float4* __restrict__ mc_all;     // [iter][0x4000]
  const RmOpts* __restrict__ opts_all;   // [iter]; uniform except .time
  float4* __restrict__ staging;          // [iter][tiles_per_part*64]
  unsigned int* __restrict__ queue;      // next local tile to hand out
  int n, iter, tile_first, tile_stride, tiles_per_part, my_tiles, resx;
  int wait_lanes;  // scheduling vote weight of continuation lanes, in 1/16
};

enum : int {
  S_IDLE = 0,     // no sample
  S_NEW,          // sample assigned, nothing computed yet
  S_RAY_STEP,     // outer march: take the next distance estimate
  S_RAY_RES,      // a distance estimate is ready for the outer march
  S_AO_STEP,      // AO: next probe or finish
  S_AO_RES,       // a distance estimate is ready for the AO loop
  S_LIGHT_STEP,   // next light or finish lighting
  S_BOUNCE_START  // start the next reflection bounce
};
enum : int { K_PRIMARY = 0, K_BOUNCE = 1, K_SHADOW = 2 };

struct WaveLane {
  // sample
  int st, id, pass, out_idx;
  float time;
  uint32_t lseed;        // light jitter seed of the sample (renderer.cl:267)
  v3 mcNormal, rd0;      // primary ray direction
  // primary hit, kept while reflections and its lighting run
  v3 h_pos, h_norm, refl;
  int h_obj, bounce;
  float h_dist;
  // reflection bounce
  v3 b_dir, b_from;
  float b_dist;
  // lighting context (level 0 = primary hit, 1 = bounce hit)
  int level, l_obj, light, ao_i;
  v3 l_pos, l_nrm, diff, spec, out, ldir;
  float ao, ao_d, att;
  uint32_t ao_seed;
  // outer march
  int rk, osteps, obj;
  float dist, maxDist;
  v3 nrm;
  // distance estimate in flight
  v3 rpos, p, delta;
  int msteps, ret;
  float g_rd, g_rc, res_d, res_c, inv_s;
  bool marching, smooth;
};

// Holds plain pointer copies only (no references to other locals), so that the
// compiler keeps every member -- and the WaveLane next to it -- in registers.
struct Leaf {
  const RmOpts* op;
  RM_DEV v3 sky(v3 d) const { return sky_of(*op, d); }
  RM_DEV v3 reflect(v3 v, v3 n) const { return reflect_of(v, n); }
  RM_DEV float schlick(float r0, float sm, v3 n, v3 view) const { return schlick_of(r0, sm, n, view); }
  RM_DEV float blinn_phong(float sm, v3 rd, v3 ld, v3 n) const { return blinn_phong_of(sm, rd, ld, n); }
  RM_DEV float box_entry(v3 p, v3 d) const { return box_entry_of(*op, p, d); }
  RM_DEV bool in_grid(int x, int y, int z) const { return in_grid_of(*op, x, y, z); }
  RM_DEV Material material(int id) const { return material_of(*op, id); }
};

struct WaveTracer {
  const WaveArgs a;
  const RmOpts& o;  // device memory, uniform
  const Leaf leaf;
  RM_DEV explicit WaveTracer(const WaveArgs& args) : a(args), o(args.opts_all[0]), leaf{args.opts_all} {}

  RM_DEV float4 tab(int pass, uint32_t seed) {
    return a.mc_all[(size_t)pass * RM_TABLE_ENTRIES + (seed & (RM_TABLE_ENTRIES - 1))];
  }
  RM_DEV v3 eye(const WaveLane& L) {  // renderer.cl:474
    return mads(V(L.mcNormal.z, L.mcNormal.x, L.mcNormal.y), o.dof, ld3(o.eyePos));
  }
  RM_DEV v3 light_at(const WaveLane& L, int i) {  // renderer.cl:263-269
    const float4 r = tab(L.pass, L.lseed);
    return mads(V(r.x, r.y, r.z), o.lightScatter, ld3(o.lightPos[i]));
  }
  // renderer.cl:275-290
  RM_DEV v3 atmosphere(const WaveLane& L, v3 ro, v3 rdir, float dist, v3 col) {
    const float fa = 1.0f - rmd::exp_det(dist * dist * -o.fogPow);
    const v3 sk = leaf.sky(rdir);
    col = V((sk.x - col.x) * fa + col.x, (sk.y - col.y) * fa + col.y, (sk.z - col.z) * fa + col.z);
    const int nl = o.numLights;
    for (int i = 0; i < nl; i++) {
      v3 lp = light_at(L, i);
      const float d = rmd::clamp_cl(dot(lp - ro, rdir), 0.0f, dist);
      lp = mads(rdir, d, ro - lp);
      const float k = o.flareAmp / dot(lp, lp);
      col = mads(ld3(o.lightColor[i]), k, col);
    }
    return col;
  }
  // (component-wise selects on values: a ternary over struct lvalues becomes a
  //  pointer select and would pin the whole lane state in scratch memory)
  RM_DEV static v3 pick(bool c, v3 x, v3 y) { return V(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z); }
  RM_DEV v3 ray_dir(const WaveLane& L) {
    return pick(L.rk == K_PRIMARY, L.rd0, pick(L.rk == K_BOUNCE, L.b_dir, L.ldir));
  }
  RM_DEV v3 ray_org(const WaveLane& L) {
    const v3 sh = mads(L.ldir, o.shadowBias, L.l_pos);
    return pick(L.rk == K_PRIMARY, eye(L), pick(L.rk == K_BOUNCE, L.b_from, sh));
  }
  RM_DEV v3 view_dir(const WaveLane& L) { return pick(L.level != 0, L.b_dir, L.rd0); }
  // per-march step vector (renderer.cl:215) and the conservative samples-per-cell factor
  RM_DEV void set_delta(WaveLane& L, v3 dir, int steps) {
    const float sf = (float)steps * 0.5f;
    L.delta = V(dir.x / sf, dir.y / sf, dir.z / sf) * ld3(o.invVoxelScale);
    const float s = fmaxf(fmaxf(__builtin_fabsf(L.delta.x) * (float)o.voxelRes[0],
                                __builtin_fabsf(L.delta.y) * (float)o.voxelRes[1]),
                          __builtin_fabsf(L.delta.z) * (float)o.voxelRes[2]);
    L.inv_s = 0.98f * __builtin_amdgcn_rcpf(fmaxf(s, 1e-6f));
  }
  // Set up one distance estimate (renderer.cl:209-218).  Either the march has to
  // run (marching = true) or the result is already known (res_d / res_c / nrm).
  RM_DEV void begin_estimate(WaveLane& L, v3 rpos, v3 dir, int steps, bool smooth, int ret) {
    const float h = rpos.y + o.groundY;
    if (h < 1e5f) { L.g_rd = h; L.g_rc = h; } else { L.g_rd = 1e5f; L.g_rc = -1.0f; }
    L.nrm = (L.g_rd < 1e5f) ? V(0.f, 1.f, 0.f) : -dir;
    L.ret = ret;
    L.rpos = rpos;
    const float t_in = leaf.box_entry(rpos, dir);
    if (t_in >= 0.0f && t_in < L.g_rd && steps > 0) {
      v3 p = rpos + ld3(o.voxelBounds);
      if (t_in > 0.0f) p = mads(dir, t_in, p);
      L.p = p * ld3(o.invVoxelScale);
      L.msteps = steps;
      L.smooth = smooth;
      L.marching = true;
    } else {
      L.res_d = L.g_rd;
      L.res_c = L.g_rc;
      L.marching = false;
      L.st = ret;
    }
  }
  RM_DEV void finish_estimate(WaveLane& L) {
    L.res_d = L.g_rd;
    L.res_c = L.g_rc;
    L.marching = false;
    L.st = L.ret;
  }

  // The hot loop: at most `budget` lookups of the fixed-step march
  // (renderer.cl:219-234) for a lane with marching == true.
  RM_DEV void march_some(WaveLane& L, int budget) {
    v3 p = L.p;
    int steps = L.msteps;
    bool done = false;
    while (budget-- > 0 && !done) {
      int cell = 0;
      const int r = walk_step(o, a.dist8, p, steps, L.delta, L.inv_s, &cell);
      if (r == 1) {
        const uint32_t w = a.surf32[cell];
        L.nrm = surf_normal(w, L.smooth);
        const v3 hit = madv(p, ld3(o.voxelBounds2), -ld3(o.voxelBounds));
        const float d = length(L.rpos - hit) - o.voxelSize;
        if (d < L.g_rd) { L.g_rd = d; L.g_rc = band_of((int)(w & 0xffu)); }
      }
      done = r != 0;
    }
    L.p = p;
    L.msteps = steps;
    if (done) finish_estimate(L);
  }

  RM_DEV void start_ray(WaveLane& L, int kind, v3 dir, float maxDist, int maxSteps) {
    L.rk = kind;
    L.dist = o.startDist;
    L.maxDist = maxDist;
    L.osteps = maxSteps;
    set_delta(L, dir, o.maxVoxelIter);
    L.st = S_RAY_STEP;
  }
  RM_DEV void start_lighting(WaveLane& L, int level) {  // renderer.cl:357, :332-337
    L.level = level;
    if (level == 0) { L.l_pos = L.h_pos; L.l_nrm = L.h_norm; L.l_obj = L.h_obj; }
    L.ao = 1.0f;
    L.ao_d = 0.0f;
    L.ao_i = 0;
    L.ao_seed = rmd::f2u(L.l_pos.x * 3183.75f + L.l_pos.y * 1831.42f + L.l_pos.z * 2945.87f +
                         L.time * 2671.918f);
    L.st = S_AO_STEP;
  }
  RM_DEV void finish_sample(WaveLane& L, v3 col) {  // renderer.cl:491; the blend is blend_kernel's
    const float e = o.exposure;
    a.staging[L.out_idx] = make_float4(col.x * e, col.y * e, col.z * e, 1.0f);
    L.st = S_IDLE;
  }

  // One continuation step of a lane that is neither marching nor idle.
  RM_DEV void advance(WaveLane& L) {
    {
      switch (L.st) {
        case S_NEW: {  // renderer.cl:467-476, :456-465
          const float t = a.opts_all[L.pass].time;
          L.time = t;
          const int resx = o.resolution[0];
          const float fx = (float)(L.id % resx), fy = (float)(L.id / resx);
          const float4 mcPos = tab(L.pass, (uint32_t)L.id * 17u + rmd::f2u(t * 3141.3862f));
          const float4 tn = tab(L.pass, (uint32_t)L.id * 37u + rmd::f2u(t * 1859.1467f));
          L.mcNormal = normalize(V(tn.x, tn.y, tn.z));
          const float px = fx + mcPos.z, py = fy + mcPos.w;
          L.lseed = rmd::f2u(px * 1957.0f + py * 2173.0f + t * 4763.742f);
          const v3 e = eye(L);
          const v3 fwd = normalize(ld3(o.targetPos) - e);
          const v3 right = normalize(cross(fwd, ld3(o.up)));
          float vx = px / (float)o.resolution[0] * o.fov - o.fov * 0.5f;
          float vy = py / (float)o.resolution[1] * o.fov - o.fov * 0.5f;
          vy *= -o.invAspect;
          const v3 upv = cross(right, fwd);
          L.rd0 = normalize(right * vx + upv * vy + fwd);
          start_ray(L, K_PRIMARY, L.rd0, o.maxDist, o.maxIter);
          break;
        }
        case S_RAY_RES:     // renderer.cl:246-250: a marched estimate came back
        case S_RAY_STEP: {  // renderer.cl:243-245
          // The outer march stays in this tight loop while its estimates need no
          // fixed-step walk (ray outside the voxel box / ground closer): those are
          // the majority of estimates and cost one slab test each.
          const v3 rd = ray_dir(L);
          const v3 ro = ray_org(L);
          bool have = L.st == S_RAY_RES;
          bool finished = false;
          for (int guard = 0; guard < 24; guard++) {
            if (have) {
              L.obj = rmd::f2i(L.res_c);
              if (__builtin_fabsf(L.res_d) <= o.eps || L.dist >= L.maxDist) { finished = true; break; }
              L.dist += L.res_d;
            }
            if (--L.osteps < 0) { finished = true; break; }
            begin_estimate(L, mads(rd, L.dist, ro), rd, o.maxVoxelIter, L.rk == K_PRIMARY, S_RAY_RES);
            if (L.marching) break;
            have = true;
          }
          if (finished) ray_done(L);
          // (guard exhausted: st == S_RAY_RES with the last result pending, resumes next step)
          break;
        }
        case S_AO_RES:     // renderer.cl:343: a probe came back
        case S_AO_STEP: {  // renderer.cl:338-342
          if (L.st == S_AO_RES) {
            L.ao *= 1.0f - rmd::fmax_cl((L.ao_d - L.res_d) * o.aoAmp / L.ao_d, 0.0f);
            L.ao_i++;
            L.st = S_AO_STEP;
          }
          if (L.ao_i <= o.aoIter && (double)L.ao > 0.01) {
            L.ao_d += o.aoStepDist;
            L.ao_seed += 37u;
            const float4 r = tab(L.pass, L.ao_seed);
            const v3 n = normalize(mads(V(r.x, r.y, r.z), 0.2f, L.l_nrm));
            const int steps = o.maxVoxelIter / 2;
            set_delta(L, n, steps);
            // (the probe's hit normal lands in L.nrm and is never read: the reference
            //  discards it too, renderer.cl:337,342)
            begin_estimate(L, mads(n, L.ao_d, L.l_pos), n, steps, false, S_AO_RES);
          } else {  // renderer.cl:358-360
            const v3 reflectCol = pick(L.level != 0, leaf.sky(leaf.reflect(L.b_dir, L.l_nrm)), L.refl);
            L.diff = leaf.sky(L.l_nrm) * L.ao;
            L.spec = reflectCol * L.ao;
            L.out = V(0.f, 0.f, 0.f);
            L.light = 0;
            L.st = S_LIGHT_STEP;
          }
          break;
        }
        case S_LIGHT_STEP: {  // renderer.cl:361-369
          if (L.light < (int)o.numLights) {
            const v3 dl = light_at(L, L.light) - L.l_pos;
            const float d2 = dot(dl, dl);
            L.att = 1.0f / d2;
            if (L.att > o.minLightAtt) {
              L.ldir = normalize(dl);
              start_ray(L, K_SHADOW, L.ldir, rmd::fmin_cl(rmd::sqrt_rn(d2) - o.shadowBias, o.maxDist),
                        o.shadowIter);
            } else {
              light_finish(L);
            }
          } else {
            lighting_done(L);
          }
          break;
        }
        case S_BOUNCE_START: {  // renderer.cl:433-435
          L.b_dir = leaf.reflect(L.b_dir, L.l_nrm);
          L.b_from = mads(L.b_dir, 0.0075f, L.l_pos);
          start_ray(L, K_BOUNCE, L.b_dir, o.maxDist, o.maxIter);
          break;
        }
        default:
          L.st = S_IDLE;
          break;
      }
    }
  }

  // renderer.cl:252-256, then what the caller of raymarch() does with the result
  RM_DEV void ray_done(WaveLane& L) {
    v3 pos = L.rpos;
    if (L.dist >= L.maxDist) {
      pos = mads(ray_dir(L), L.dist, ray_org(L));
      L.obj = -1;
      L.dist = 1000.0f;
    }
    if (L.rk == K_SHADOW) {  // renderer.cl:300, :370-374
      const float sh = rmd::step_cl(L.maxDist, L.dist);
      if (sh > 0.0f) {
        const Material m = leaf.material(L.l_obj);
        const v3 raydir = view_dir(L);
        const v3 inc = (ld3(o.lightColor[L.light]) * sh) * L.att;
        L.diff = L.diff + inc * rmd::fmax_cl(0.0f, dot(L.ldir, L.l_nrm));
        L.spec = L.spec + inc * leaf.blinn_phong(m.smoothness, raydir, L.ldir, L.l_nrm);
      }
      light_finish(L);
    } else if (L.rk == K_PRIMARY) {  // renderer.cl:415-441
      if (L.dist >= o.maxDist) {
        finish_sample(L, atmosphere(L, eye(L), L.rd0, L.dist, leaf.sky(L.rd0)));
        return;
      }
      const Material m = leaf.material(L.obj);
      const float k = 1.0f / (m.smoothness * 200.0f + 5.0f);
      L.h_norm = mads(L.mcNormal, k, L.nrm);
      L.h_pos = pos;
      L.h_obj = L.obj;
      L.h_dist = L.dist;
      L.refl = V(0.f, 0.f, 0.f);
      if (m.r0 > 0.0f && o.reflectIter > 0) {
        L.bounce = 0;
        L.l_pos = L.h_pos;
        L.l_nrm = L.h_norm;
        L.b_dir = L.rd0;
        L.st = S_BOUNCE_START;
      } else {
        L.refl = leaf.sky(leaf.reflect(L.rd0, L.h_norm));
        start_lighting(L, 0);
      }
    } else {  // K_BOUNCE: renderer.cl:389-404, :436
      L.b_dist = L.dist;
      L.l_obj = L.obj;
      if (L.obj < 0) {
        L.refl = L.refl + atmosphere(L, L.b_from, L.b_dir, L.b_dist, leaf.sky(L.b_dir));
        start_lighting(L, 0);
      } else {
        L.l_pos = pos;
        L.l_nrm = L.nrm;
        start_lighting(L, 1);
      }
    }
  }
  // renderer.cl:376-378
  RM_DEV void light_finish(WaveLane& L) {
    const Material m = leaf.material(L.l_obj);
    const v3 raydir = view_dir(L);
    L.diff = L.diff * m.albedo;
    L.out = L.out + mixs(L.diff, L.spec, leaf.schlick(m.r0, m.smoothness, L.l_nrm, raydir));
    L.light++;
    L.st = S_LIGHT_STEP;
  }
  // renderer.cl:380, then :404 / :435-437 (bounce) or :444 (primary)
  RM_DEV void lighting_done(WaveLane& L) {
    const float fl = (float)o.numLights;
    const v3 col = V(L.out.x / fl, L.out.y / fl, L.out.z / fl);
    if (L.level) {
      L.refl = L.refl + atmosphere(L, L.b_from, L.b_dir, L.b_dist, col);
      L.bounce++;
      const bool more = L.bounce < o.reflectIter && !((double)leaf.material(L.l_obj).r0 < 0.001);
      if (more) L.st = S_BOUNCE_START;
      else start_lighting(L, 0);
    } else {
      finish_sample(L, atmosphere(L, eye(L), L.rd0, L.h_dist, col));
    }
  }
};

}  // namespace rmk
