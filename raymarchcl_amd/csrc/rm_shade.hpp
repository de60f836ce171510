// rm_shade.hpp -- per-sample ray march + shading for gfx950, written from the
// behaviour of the reference device functions
// (/root/reference/resources/renderer.cl:142-476; the function each routine
// replaces is cited next to it).  Scalar float32 throughout, every expression
// in the reference's evaluation order; the value of the reference's math built-ins, of its `/` and of the
// a*b+c it spells inside one expression (which ITS compiler may contract) comes from the arithmetic
// contract M (rm_math.hpp: OpenCL CPU device / this GPU, strict or default build of the reference).
//
// One lane owns one sample from camera ray to final colour (shade()); in the frame kernel
// the AO probes and shadow rays of a wavefront's hits are traced by all its lanes
// (shade_wave()).  rm_kernels.hip wraps both in the kernels.
#pragma once
#include "rm_math.hpp"
#include "rm_opts.h"

namespace rmk {

// Optional device-side event counters (algorithmic bytes for the roofline).
struct Counters {
  unsigned long long vox_reads, mc_reads, rays, dts_calls, march_steps, ao_calls, primary_hits,
      oob_material;
};

struct Material { v3 albedo; float r0, smoothness; };

// Everything a sample needs that is uniform across the launch.
struct Scene {
  const uint8_t* __restrict__ vox;
  const float4* __restrict__ mc;
  const RmOpts* __restrict__ o;
  const uint8_t* __restrict__ dist;   // rm_accel.hip dist8, or nullptr
  const uint32_t* __restrict__ surf;  // rm_accel.hip surf32, or nullptr
  unsigned long long oct_stride = 0;  // > 0: 8 directional tables follow dist8 (rm_accel.hip oct8)
  const float* __restrict__ sdf = nullptr;  // quality mode: the distance field, one float4 xy-face per cell (Tracer<.., SDFM = true>)
  unsigned log2res = 0;   // LAYOUT 2 (walk_step): edge of the cubic grid = 1 << log2res
};

// ---- leaf routines; `o` points at the option record in device memory ----

// materials[id] by byte offset, defined for every id (see oracle/rm_restate.c);
// *oob is set when the index falls outside the record (undefined in the reference)
RM_DEV Material material_of(const RmOpts& o, int id, bool* oob = nullptr) {
  Material m{V(0.f, 0.f, 0.f), 0.f, 0.f};
  const int off = (int)offsetof(RmOpts, materials) + (int)sizeof(RmMaterial) * id;
  if (id < -13 || id > 3) {
    if (oob) *oob = true;
    return m;
  }
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(&o) + off);
  m.albedo = V(p[0], p[1], p[2]);
  m.r0 = p[4];
  m.smoothness = p[5];
  return m;
}
// slab test: renderer.cl:153-161
template <class M>
RM_DEV float box_entry_of(const RmOpts& o, v3 p, v3 d) {
  d = M::slab_dir(d);
  const float lox = M::div(o.voxelBoundsMin[0] - p.x, d.x), loy = M::div(o.voxelBoundsMin[1] - p.y, d.y),
              loz = M::div(o.voxelBoundsMin[2] - p.z, d.z);
  const float hix = M::div(o.voxelBoundsMax[0] - p.x, d.x), hiy = M::div(o.voxelBoundsMax[1] - p.y, d.y),
              hiz = M::div(o.voxelBoundsMax[2] - p.z, d.z);
  const float nx = M::fmin(hix, lox), ny = M::fmin(hiy, loy), nz = M::fmin(hiz, loz);
  const float a = M::fmax(M::fmax(nx, 0.0f), M::fmax(ny, nz));
  const float fx = M::fmax(hix, lox), fy = M::fmax(hiy, loy), fz = M::fmax(hiz, loz);
  const float b = M::fmin(fx, M::fmin(fy, fz));
  return b > a ? a : -1.0f;
}
RM_DEV bool in_grid_of(const RmOpts& o, int qx, int qy, int qz) {  // 0 <= q < res per axis
  return ((unsigned)qz < (unsigned)o.voxelRes[2]) & ((unsigned)qy < (unsigned)o.voxelRes[1]) &
         ((unsigned)qx < (unsigned)o.voxelRes[0]);
}
// renderer.cl:259-261
template <class M>
RM_DEV v3 sky_of(const RmOpts& o, v3 dir) {
  return M::mixs(ld3(o.skyColor1), ld3(o.skyColor2), M::fuse(dir.y, 0.5f, 0.5f));
}
// renderer.cl:271-273
template <class M>
RM_DEV v3 reflect_of(v3 v, v3 n) {
  const float k = 2.0f * M::dot(v, n);
  return M::nfuse3(n, k, v);  // v - 2.0f * dot(v, n) * n
}
// renderer.cl:304-311
template <class M>
RM_DEV float schlick_of(float r0, float smooth, v3 n, v3 view) {
  const float d = M::clamp(1.0f - M::dot(n, -view), 0.0f, 1.0f);
  if (d > 0.0f) {
    const float d2 = d * d;
    return M::mad(1.0f - r0, smooth * d2 * d2 * d, r0);
  }
  return 0.0f;
}
// renderer.cl:317-325
template <class M>
RM_DEV float blinn_phong_of(float smooth, v3 raydir, v3 ldir, v3 n) {
  const float nh = M::dot(M::normalize(ldir - raydir), n);
  if (nh > 0.0f) {
    const float sp = M::exp2(M::mad(6.0f, smooth, 4.0f));
    return M::pow(nh, sp) * (sp + 2.0f) * 0.125f;
  }
  return 0.0f;
}
// decode of a surf32 word (rm_accel.hip) into the hit normal and material code
template <class M>
RM_DEV v3 surf_normal(uint32_t w, bool smooth) {
  if (smooth)
    return M::normalize(V((float)((int)((w >> 8) & 63u) - 32), (float)((int)((w >> 14) & 63u) - 32),
                       (float)((int)((w >> 20) & 63u) - 32)));
  return M::normalize(V(-(float)((int)((w >> 26) & 3u) - 1), -(float)((int)((w >> 28) & 3u) - 1),
                        -(float)((int)((w >> 30) & 3u) - 1)));
}
RM_DEV float band_of(int v) { return v < 168 ? (v < 84 ? 1.0f : 2.0f) : 3.0f; }  // renderer.cl:205-207

// One sample of the accelerated fixed-step walk (renderer.cl:219-234): look the
// sample at p up in the skip table and move on.  `steps` = samples left including this one.
// Returns 0: keep walking, 1: this sample is a hit (*cell_out = its cell, p kept),
// 2: the walk ends without a hit (out of samples / left the grid / cannot reach
// anything in the samples left).
//
// The samples that are skipped still advance p by the reference's sequential adds,
// so the positions of all later samples -- and the hit position -- are bit-identical.
// Skip length: a table value d at cell q means every cell within Chebyshev distance d-1
// of q (dist8) / of the cube of edge d ahead of the walk (oct8) is empty and inside the grid.
// Sample k lies <= k*s cells from sample 0 along the fastest axis (s = cells per sample), so
// its cell differs from q by at most floor(k*s + eps) + 1; samples 1 .. j-1 are therefore
// certainly empty for
//     j = 1 + floor(0.98 * (d-1) / s) = floor(d * inv_s + (1 - inv_s)),  inv_s = 0.98 / s
// (the 2 % absorb the <= 0.01 cell of accumulated rounding drift, the rounding of p*res and of
// the one fma that evaluates it; max(., 1) catches the rounding at d = 1).  Cells on a face of
// the grid hold d = 1 -- also the cell a slightly negative coordinate truncates to
// (renderer.cl:165) when the walk heads outward -- so no sample outside the grid is skipped.
//
// LAYOUT of the nine tables (chosen by the host per volume):
//   0  row-major, 64-bit table offsets                                   (any grid)
//   1  8x4x4-cell bricks of 128 bytes = one cache line each, 64-bit offsets: volumes whose
//      tables exceed the Infinity Cache (a ray's consecutive fetches and the lanes of a
//      wavefront share lines instead of touching a new row segment per fetch)
//   2  row-major, cubic power-of-two grid, all tables below 4 GiB: the cell index is two
//      shift-ors, the bounds test one compare of the or-ed coordinates, and the table is read
//      through a buffer descriptor with a 32-bit offset (no 64-bit address arithmetic per
//      fetch; out-of-range offsets read 0, so the fetch needs no guard)
//   3  the bricks of layout 1 with the arithmetic of layout 2, on the 512^3 grid (nine tables = 1.2 GB: beyond the
//      Infinity Cache, where the locality of bricks pays, and below the 4 GiB one buffer descriptor reaches):
//      positions in cell units, one bounds compare, brick number and byte within the brick by shifts and ors with
//      the grid edge as a compile-time constant (three scalar registers fewer), 32-bit buffer offset
//   5  layout 2 for the 256^3 grid with the edge compiled in (shift counts, bound and scale as immediates: the
//      allocator keeps 36 instead of 42 values in scratch)
//   4  the same for the 1024^3 grid: its nine tables are 9 GiB, beyond any buffer descriptor (the hardware forms
//      index * stride in 32 bits), so the byte is fetched by a plain 64-bit address = table number << 30 | brick
//      offset, guarded by the bounds compare
constexpr unsigned kLog2Res3 = 9;  // LAYOUT 3: the grid is 512^3
constexpr unsigned kLog2Res4 = 10; // LAYOUT 4: the grid is 1024^3 (the same bricks behind 64-bit addresses: 9 GiB of tables)
constexpr unsigned kLog2Res5 = 8;  // LAYOUT 5: layout 2 on the 256^3 grid, edge compiled in
constexpr bool bricks_by_shifts(int layout) { return layout == 3 || layout == 4; }
// grid edge (log2) that a layout has compiled in; 0 = read from the table descriptor at run time
constexpr unsigned fixed_log2(int layout) {
  return layout == 3 ? kLog2Res3 : (layout == 4 ? kLog2Res4 : (layout == 5 ? kLog2Res5 : 0u));
}
struct WalkTab {
  const uint8_t* __restrict__ dist8;   // table 0; tables 1..8 follow at oct_stride
  __amdgpu_buffer_rsrc_t rsrc;         // LAYOUT 2, 3, 5: all nine tables as one buffer
  unsigned res, sh;                    // LAYOUT 2: edge of the grid = 1 << sh
  float fres;
};
// skip = the reference's own adds (renderer.cl:233) without the fetches.  (A closed
// form p + j*D, exact while p stays inside one binade, was measured slower: three
// adds per skipped sample are cheaper than its bookkeeping and failure path.)
// (on average 5 samples are advanced per fetch: the adds are unrolled by four so that
//  the loop bookkeeping does not cost more than the adds themselves)
RM_DEV void skip_samples(v3& p, v3 delta, int j) {
  p = p + delta;
  int k = j - 1;
  while (k >= 4) {
    p = p + delta;
    p = p + delta;
    p = p + delta;
    p = p + delta;
    k -= 4;
  }
  if (k & 2) {
    p = p + delta;
    p = p + delta;
  }
  if (k & 1) p = p + delta;
}
template <class M, int LAYOUT>
RM_DEV int walk_step(const RmOpts& o, const WalkTab& tab, v3& p, int& steps, v3 delta, float inv_s, float c0,
                     int* cell_out, unsigned long long table_off) {
  int d, j;
  unsigned cell;
  if (LAYOUT >= 2) {
    // (M::cell: the bare conversion instruction; scene_distance has applied M::walk_guard)
    // p and delta are in CELL units in this layout: the reference's p * res (renderer.cl:165) with res a
    // power of two is an exact scaling that commutes with the rounding of every add (an add whose result
    // is subnormal is exact either way), so scene_distance scales the first sample and the step once and
    // the walk's p is, bit for bit, 2^k times the reference's at every sample
    const int qx = M::cell(p.x), qy = M::cell(p.y), qz = M::cell(p.z);
    const unsigned sh = fixed_log2(LAYOUT) ? fixed_log2(LAYOUT) : tab.sh, res = fixed_log2(LAYOUT) ? (1u << fixed_log2(LAYOUT)) : tab.res;
    const bool ok = ((((unsigned)qx | (unsigned)qy) | (unsigned)qz) < res) & (steps > 0);  // renderer.cl:219, :221
    cell = ((((unsigned)qz << sh) | (unsigned)qy) << sh) | (unsigned)qx;  // (surf32 stays row-major)
    unsigned at = cell;
    if (bricks_by_shifts(LAYOUT)) {  // 8x4x4-cell bricks of 128 bytes, x fastest inside and between bricks (rm_accel.hip tab_index)
      const unsigned brick = (((((unsigned)qz >> 2) << (sh - 2u)) | ((unsigned)qy >> 2)) << (sh - 3u)) | ((unsigned)qx >> 3);
      const unsigned within = (((((unsigned)qz & 3u) << 2) | ((unsigned)qy & 3u)) << 3) | ((unsigned)qx & 7u);
      at = (brick << 7) | within;
    }
    if (LAYOUT == 4)  // (no descriptor reaches 9 GiB: a plain load, which needs its guard; table_off = table number)
      d = ok ? (int)tab.dist8[(table_off << (3u * kLog2Res4)) + at] : 0;
    else
      d = (int)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(tab.rsrc, at + (unsigned)table_off, 0, 0);
    j = max((int)__builtin_fmaf((float)d, inv_s, c0), 1);
    *cell_out = (int)cell;
    // one decision per sample: hit / go on / end
    const bool hit = ok & (d == 0);
    const bool go = ok & (d != 0) & (j < steps);
    if (!go) return hit ? 1 : 2;
  } else {
    const int qx = M::cell(p.x * (float)o.voxelRes[0]);
    const int qy = M::cell(p.y * (float)o.voxelRes[1]);
    const int qz = M::cell(p.z * (float)o.voxelRes[2]);
    if (!(in_grid_of(o, qx, qy, qz) & (steps > 0))) return 2;  // renderer.cl:219, :221
    // (qz*ry + qy)*rx + qx with 24-bit multiplies (full rate; the host only enables the
    // derived structures when ry*rz < 2^24 and rx < 2^24); the table offset is 64-bit
    if (LAYOUT == 1) {
      const unsigned nbx = ((unsigned)o.voxelRes[0] + 7u) >> 3, nby = ((unsigned)o.voxelRes[1] + 3u) >> 2;
      const unsigned brick = __umul24(__umul24((unsigned)qz >> 2, nby) + ((unsigned)qy >> 2), nbx) + ((unsigned)qx >> 3);
      const unsigned within = ((((unsigned)qz & 3u) << 2 | ((unsigned)qy & 3u)) << 3) | ((unsigned)qx & 7u);
      cell = 0;
      // (64-bit: the bricks of a 1024^3 table alone are 1 GiB; 9 tables follow each other)
      d = tab.dist8[(((unsigned long long)brick << 7) | within) + table_off];
    } else {
      cell = __umul24(__umul24((unsigned)qz, (unsigned)o.voxelRes[1]) + (unsigned)qy,
                      (unsigned)o.voxelRes[0]) + (unsigned)qx;
      d = tab.dist8[cell + table_off];
    }
    if (d == 0) {
      if (LAYOUT == 1)  // surf32 stays row-major
        cell = __umul24(__umul24((unsigned)qz, (unsigned)o.voxelRes[1]) + (unsigned)qy, (unsigned)o.voxelRes[0]) +
               (unsigned)qx;
      *cell_out = (int)cell;
      return 1;
    }
    j = max((int)__builtin_fmaf((float)d, inv_s, c0), 1);
    if (j >= steps) return 2;  // no sample left that could hit anything
  }
  skip_samples(p, delta, j);
  steps -= j;
  return 0;
}

// SDFM: QUALITY MODE -- not the reference's algorithm: distance estimates
// come from a trilinearly sampled float field, normals from its gradient, shadows are soft.
// M: the arithmetic contract (rm_math.hpp ArithOf): MathX86<0> OpenCL CPU device, MathX86<1> the same with the GPU lowering
// of the seed casts, MathOcl / MathOclDef ROCm's OpenCL library on this GPU as the strict / default build of the reference uses it
template <bool COUNT, bool ACCEL = false, bool SDFM = false, int LAYOUT = 0, class M = MathX86<0>>
struct Tracer {
  // the reference's built-ins under the contract (unqualified calls below resolve to these)
  RM_DEV static float dot(v3 a, v3 b) { return M::dot(a, b); }
  RM_DEV static v3 cross(v3 a, v3 b) { return M::cross(a, b); }
  RM_DEV static v3 normalize(v3 a) { return M::normalize(a); }
  RM_DEV static float length(v3 a) { return M::length(a); }
  RM_DEV static v3 mads(v3 a, float s, v3 c) { return M::mads(a, s, c); }
  RM_DEV static v3 madv(v3 a, v3 b, v3 c) { return M::madv(a, b, c); }
  RM_DEV static v3 mixs(v3 a, v3 b, float t) { return M::mixs(a, b, t); }
  static_assert(!(COUNT && ACCEL), "event counts are defined on the reference algorithm");
  static_assert(!(SDFM && (COUNT || ACCEL)), "the quality mode has no counters and no derived tables");
  const Scene& sc;
  // The pass a lane works on: its scatter table and its record's .time.  Defaults
  // to the scene's (uniform); set_pass() makes them per-lane so that one wavefront
  // can hold several passes of the same pixels (records equal except .time).
  const float4* mc_;
  float time_;
  Counters cnt;  // per-lane, only touched when COUNT
  // Wave-shared scratch in LDS (kWaveLdsFloats floats of ONE wavefront), or nullptr: lets the
  // lanes of a wavefront hand AO probes and shadow rays to each other (shade_wave()).
  float* lds_ = nullptr;
  WalkTab tab_;  // the skip tables as walk_step reads them (uniform)
  RM_DEV explicit Tracer(const Scene& s) : sc(s), mc_(s.mc), time_(s.o->time), cnt{} {
    tab_.dist8 = s.dist;
    tab_.sh = s.log2res;
    tab_.res = 1u << s.log2res;
    tab_.fres = (float)(1u << s.log2res);
    if (LAYOUT == 2 || LAYOUT == 3 || LAYOUT == 5) {
      const unsigned long long bytes = (s.oct_stride ? 9ull : 1ull) << (3u * s.log2res);  // < 4 GiB (host)
      tab_.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(s.dist), 0, (int)(unsigned)bytes, 0x00020000);
    }
  }
  RM_DEV void set_pass(const float4* table_of_pass, float time_of_pass) {
    mc_ = table_of_pass;
    time_ = time_of_pass;
  }

  // the (uint) cast of a seed expression (renderer.cl:267, 334, 471, 472): undefined outside
  // [0, 2^32); lowered as the contract's device lowers it
  RM_DEV uint32_t seed_of(float x) { return M::seed(x); }
  // scatter table lookup: renderer.cl:142-144
  RM_DEV float4 table(uint32_t seed) {
    if (COUNT) cnt.mc_reads++;
    return mc_[seed & (RM_TABLE_ENTRIES - 1)];
  }

  RM_DEV Material material(int id) {
    bool oob = false;
    const Material m = material_of(*sc.o, id, &oob);
    if (COUNT && oob) cnt.oob_material++;
    return m;
  }
  RM_DEV float box_entry(v3 p, v3 d) { return box_entry_of<M>(*sc.o, p, d); }

  RM_DEV bool in_grid(int qx, int qy, int qz) { return in_grid_of(*sc.o, qx, qy, qz); }
  // binary occupancy: renderer.cl:172-178
  RM_DEV float solid(int qx, int qy, int qz) {
    const RmOpts& o = *sc.o;
    if (!in_grid(qx, qy, qz)) return 0.0f;
    if (COUNT) cnt.vox_reads++;
    return M::step((float)o.isoVal, (float)sc.vox[qz * o.voxelRes[3] + qy * o.voxelRes[0] + qx]);
  }
  // negated central difference: renderer.cl:180-188
  RM_DEV v3 cell_gradient(int qx, int qy, int qz) {
    const float gx = solid(qx + 1, qy, qz) - solid(qx - 1, qy, qz);
    const float gy = solid(qx, qy + 1, qz) - solid(qx, qy - 1, qz);
    const float gz = solid(qx, qy, qz + 1) - solid(qx, qy, qz - 1);
    return V(-gx, -gy, -gz);
  }
  // 3x3x3 sum of gradients of solid cells: renderer.cl:190-203
  RM_DEV v3 smooth_gradient(int qx, int qy, int qz) {
    v3 n = V(0.f, 0.f, 0.f);
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++)
          if (solid(qx + dx, qy + dy, qz + dz) > 0.0f)
            n = n + cell_gradient(qx + dx, qy + dy, qz + dz);
    return normalize(n);
  }

  // ---- QUALITY MODE (mirrors oracle/rm_restate.c sdf_* operation by operation) ----
  // field value at a position inside (or on) the clip box: trilinear over cell centres
  RM_DEV float sdf_sample(v3 q) {
    const RmOpts& o = *sc.o;
    const int rx = o.voxelRes[0], ry = o.voxelRes[1], rz = o.voxelRes[2];
    const float ux = M::clamp((q.x + o.voxelBounds[0]) * o.invVoxelScale[0] * (float)rx - 0.5f, 0.0f, (float)(rx - 1));
    const float uy = M::clamp((q.y + o.voxelBounds[1]) * o.invVoxelScale[1] * (float)ry - 0.5f, 0.0f, (float)(ry - 1));
    const float uz = M::clamp((q.z + o.voxelBounds[2]) * o.invVoxelScale[2] * (float)rz - 0.5f, 0.0f, (float)(rz - 1));
    int ix = (int)ux, iy = (int)uy, iz = (int)uz;  // u >= 0: truncation = floor
    ix = min(ix, rx - 2); iy = min(iy, ry - 2); iz = min(iz, rz - 2);
    ix = max(ix, 0); iy = max(iy, 0); iz = max(iz, 0);
    const float fx = ux - (float)ix, fy = uy - (float)iy, fz = uz - (float)iz;
    // The host admits fields of 2 .. 4096 cells per axis (rm_set_sdf_volume), so the upper neighbour of the clamped
    // base cell always exists and a cell index fits 32 bits.  sc.sdf holds one float4 per cell: the four field values
    // of its xy-face (rm_accel.hip sdf_quads_kernel) -- TWO 16-byte loads per sample (this layer and the next)
    // instead of eight scalar ones: the kernel is bound by the L1's tag lookups, one per lane and load for rays
    // this incoherent.  Same values, same arithmetic as the row-major reads of the restatement.
    const float4* __restrict__ g = reinterpret_cast<const float4*>(sc.sdf);
    const unsigned r00 = __umul24(__umul24((unsigned)iz, (unsigned)ry) + (unsigned)iy, (unsigned)rx) + (unsigned)ix;
    const float4 t0 = g[r00], t1 = g[r00 + __umul24((unsigned)ry, (unsigned)rx)];
    const float a00 = t0.x + (t0.y - t0.x) * fx;
    const float a10 = t0.z + (t0.w - t0.z) * fx;
    const float a01 = t1.x + (t1.y - t1.x) * fx;
    const float a11 = t1.z + (t1.w - t1.z) * fx;
    const float b0 = a00 + (a10 - a00) * fy;
    const float b1 = a01 + (a11 - a01) * fy;
    return b0 + (b1 - b0) * fz;
  }
  // distance to the field's zero set from anywhere: outside the clip box the distance to
  // the box is added to the value at the nearest point of the box
  RM_DEV float sdf_volume(v3 p) {
    const RmOpts& o = *sc.o;
    const v3 q = V(M::clamp(p.x, o.voxelBoundsMin[0], o.voxelBoundsMax[0]),
                   M::clamp(p.y, o.voxelBoundsMin[1], o.voxelBoundsMax[1]),
                   M::clamp(p.z, o.voxelBoundsMin[2], o.voxelBoundsMax[2]));
    return sdf_sample(q) + length(p - q);
  }
  // gradient of the field at a position (central differences over one voxelSize)
  RM_DEV v3 sdf_gradient(v3 rpos) {
    const float e = sc.o->voxelSize;
    const v3 g = V(sdf_volume(V(rpos.x + e, rpos.y, rpos.z)) - sdf_volume(V(rpos.x - e, rpos.y, rpos.z)),
                   sdf_volume(V(rpos.x, rpos.y + e, rpos.z)) - sdf_volume(V(rpos.x, rpos.y - e, rpos.z)),
                   sdf_volume(V(rpos.x, rpos.y, rpos.z + e)) - sdf_volume(V(rpos.x, rpos.y, rpos.z - e)));
    return normalize(g);
  }
  // grad_later: nullptr = the normal of an estimate that is close enough to be the hit is computed here;
  // otherwise *grad_later says that it IS the gradient at rpos and leaves computing it to the caller (a march
  // keeps only its LAST estimate's normal, renderer.cl:244-246: six more field samples once per march instead
  // of in every turn in which some lane of the wavefront is within 2 eps)
  RM_DEV void scene_distance_sdf(v3 rpos, v3 dir, float& dist, float& code, v3& nrm, bool* grad_later = nullptr) {
    const RmOpts& o = *sc.o;
    const float h = rpos.y + o.groundY;
    float rd, rc;
    if (h < 1e5f) { rd = h; rc = h; } else { rd = 1e5f; rc = -1.0f; }
    nrm = (rd < 1e5f) ? V(0.f, 1.f, 0.f) : -dir;
    const float dv = sdf_volume(rpos);
    bool later = false;
    if (dv < rd) {
      rd = dv;
      rc = 1.0f;
      nrm = -dir;
      if (dv <= o.eps * 2.0f) {  // close enough to be the hit: gradient by central differences
        if (grad_later) later = true;
        else nrm = sdf_gradient(rpos);
      }
    }
    if (grad_later) *grad_later = later;
    dist = rd;
    code = rc;
  }
  // penumbra estimate along a light ray: min over the march of k * clearance / distance
  RM_DEV float soft_shadow_sdf(v3 p, v3 ldir, float lmax) {
    const RmOpts& o = *sc.o;
    const float k = 1.0f / M::fmax(o.lightScatter, 0.01f);
    float res = 1.0f;
    float t = 0.0f;
    for (int i = 0; i < o.shadowIter; i++) {
      const v3 q = mads(ldir, t, p);
      const float h = q.y + o.groundY;
      const float d = M::fmin(h < 1e5f ? h : 1e5f, sdf_volume(q));
      if (d <= o.eps * 0.5f) return 0.0f;
      res = M::fmin(res, k * d / (t + o.shadowBias));
      t += M::fmax(d, o.eps);
      if (t >= lmax) break;
    }
    return M::clamp(res, 0.0f, 1.0f);
  }

  // ---- walks that only have to look as far as their result can depend on ----
  // A caller that uses nothing but the DISTANCE of an estimate (AO probes, shadow marches: no
  // normal, no material) gets min(ground term g, |rpos - hit| - voxelSize): a hit farther than
  // g + voxelSize changes nothing.  More: an AO probe at distance d contributes
  // 1 - max((d - sd)*aoAmp/d, 0) (renderer.cl:343), exactly 1 for every sd >= d (aoAmp >= 0,
  // d > 0); a shadow march (renderer.cl:292-301) only reports whether it passes maxDist, and an
  // estimate larger than the remaining distance + eps ends it the same way whatever its value.
  // So such a walk only needs the samples within `reach` (+ voxelSize) of its start: sample k
  // lies at least k * (smallest world step) away, which gives the sample count below; the +2
  // (a hit beyond the limit is at least one whole step, >= 0.01 world units, past reach +
  // voxelSize) covers the roundings of the reference's own distance evaluation, ~1e-6.  The
  // samples that ARE walked are the reference's, so a hit within reach is bit-identical.
  // (dir_len: length of the walk direction; reflected directions are not unit vectors because
  //  the reference reflects about an un-normalised normal, renderer.cl:420, :434)
  // samples per world unit of a walk, rounded UP by 0.2 % (hardware reciprocal: the limit is a
  // bound, only its direction of error matters); 0 = unusable
  RM_DEV float samples_per_unit(int steps, float dir_len = 1.0f) {
    const RmOpts& o = *sc.o;
    // world advance per sample >= |dir| * min_axis(invVoxelScale * voxelBounds2) / (steps/2)
    const float sc_min = fminf(fminf(__builtin_fabsf(o.invVoxelScale[0] * o.voxelBounds2[0]),
                                     __builtin_fabsf(o.invVoxelScale[1] * o.voxelBounds2[1])),
                               __builtin_fabsf(o.invVoxelScale[2] * o.voxelBounds2[2]));
    const float adv = sc_min * dir_len;
    if (!(adv > 1e-6f)) return 0.0f;
    return 1.002f * ((float)steps * 0.5f) * __builtin_amdgcn_rcpf(adv);
  }
  RM_DEV int walk_limit_from(float reach, float spu) {
    const RmOpts& o = *sc.o;
    const float k = (fmaxf(reach, 0.0f) + __builtin_fabsf(o.voxelSize)) * spu + 2.0f;
    return ((spu > 0.0f) & (k < 1e9f)) ? (int)k : 0x7fffffff;  // (NaN reach: no limit)
  }
  RM_DEV int walk_limit_for(float reach, int steps, float dir_len = 1.0f) {
    return walk_limit_from(reach, samples_per_unit(steps, dir_len));
  }
  // AO probe at distance d whose start is g above the ground
  RM_DEV int ao_walk_limit(float d, float g, int steps) {
    const RmOpts& o = *sc.o;
    if (!(o.aoAmp >= 0.0f) || !(d > 0.0f)) return walk_limit_for(g, steps);
    return walk_limit_for(fminf(d, g), steps);
  }

  // distance estimate: renderer.cl:209-237
  // known_inside: the caller has established that rpos lies inside the clip box by a margin (no caller does since
  // round 5: the test below decides); the reference's slab test then returns exactly +0.
  // walk_limit: the caller has no use for a hit beyond that many samples (ao_walk_limit);
  // accelerated path only -- the step vector still comes from `steps`.
  RM_DEV void scene_distance(v3 rpos, v3 dir, int steps, bool smooth, float& dist, float& code,
                             v3& nrm, bool known_inside = false, int walk_limit = 0x7fffffff,
                             bool* cut = nullptr) {
    const RmOpts& o = *sc.o;
    if (SDFM) {  // (`cut` doubles as "the normal is the gradient, not computed yet")
      scene_distance_sdf(rpos, dir, dist, code, nrm, cut);
      return;
    }
    if (COUNT) cnt.dts_calls++;
    const float h = rpos.y + o.groundY;
    float rd, rc;
    if (h < 1e5f) { rd = h; rc = h; } else { rd = 1e5f; rc = -1.0f; }
    nrm = (rd < 1e5f) ? V(0.f, 1.f, 0.f) : -dir;
    // A position inside the clip box by a margin makes every slab pair (neg, pos)/d:
    // all three entry parameters are negative (or -inf), all exits positive, and the
    // reference's slab test returns max(.., 0) = exactly +0 -- without six divisions.
    // (AO probes start on a surface inside the box and almost always qualify.)
    if (ACCEL && !known_inside) {
      const float m = 1e-4f;
      known_inside = (rpos.x - o.voxelBoundsMin[0] > m) & (o.voxelBoundsMax[0] - rpos.x > m) &
                     (rpos.y - o.voxelBoundsMin[1] > m) & (o.voxelBoundsMax[1] - rpos.y > m) &
                     (rpos.z - o.voxelBoundsMin[2] > m) & (o.voxelBoundsMax[2] - rpos.z > m);
    }
    const float t_in = known_inside ? 0.0f : box_entry(rpos, dir);
    if (t_in >= 0.0f && t_in < rd) {
      const float sf = (float)steps * 0.5f;
      const v3 ivs = ld3(o.invVoxelScale);
      v3 delta;
      if (ACCEL && M::kExactDiv) {  // (the IEEE quotients in three instructions each)
        const rmd::Divisor by_sf = rmd::make_divisor(sf);
        delta = V(rmd::div_by(dir.x, by_sf), rmd::div_by(dir.y, by_sf), rmd::div_by(dir.z, by_sf)) * ivs;
      } else {
        delta = V(M::div(dir.x, sf), M::div(dir.y, sf), M::div(dir.z, sf)) * ivs;
      }
      v3 p = rpos + ld3(o.voxelBounds);
      if (t_in > 0.0f) p = mads(dir, t_in, p);
      p = p * ivs;
      const float frx = (float)o.voxelRes[0], fry = (float)o.voxelRes[1], frz = (float)o.voxelRes[2];
      const int iso = o.isoVal;
      if (ACCEL) {
        const bool limited = walk_limit < steps;
        if (limited) steps = walk_limit;
        M::walk_guard(p, delta, steps);  // (device contract: a NaN operand ends the walk where the library conversion would)
        // cells per sample along the fastest axis, padded: bounds how many samples
        // certainly stay inside the empty neighbourhood dist8 reports
        const float fres = fixed_log2(LAYOUT) ? (float)(1u << fixed_log2(LAYOUT)) : tab_.fres;
        if (LAYOUT >= 2) {  // cell units (walk_step): exact, the grid edge is a power of two
          p = p * fres;
          delta = delta * fres;
        }
        const float s = LAYOUT >= 2 ? fmaxf(fmaxf(__builtin_fabsf(delta.x), __builtin_fabsf(delta.y)), __builtin_fabsf(delta.z))
                                    : fmaxf(fmaxf(__builtin_fabsf(delta.x) * frx, __builtin_fabsf(delta.y) * fry),
                                            __builtin_fabsf(delta.z) * frz);
        const float inv_s = 0.98f * __builtin_amdgcn_rcpf(fmaxf(s, 1e-6f));
        const float c0 = 1.0f - inv_s;
        // directional table of this walk (a walk never moves against the signs of delta)
        unsigned long long table_off = 0;  // 64-bit: nine 1024^3 tables span 9 GiB
        if (sc.oct_stride) {
          const unsigned int oct = (delta.x < 0.0f ? 1u : 0u) | (delta.y < 0.0f ? 2u : 0u) | (delta.z < 0.0f ? 4u : 0u);
          table_off = LAYOUT == 4 ? (unsigned long long)(oct + 1u) : (oct + 1u) * sc.oct_stride;
        }
        (void)s;
        // the loop holds nothing but the walk: a lane that finds its hit waits for the
        // others and all hits are then evaluated together (inside the loop the compiler
        // runs the hit code once per trip in which any lane finishes)
        int cell = 0, r;
        do {
          r = walk_step<M, LAYOUT>(o, tab_, p, steps, delta, inv_s, c0, &cell, table_off);
        } while (r == 0);
        if (cut) *cut = limited & (r != 1);  // ended without a hit, possibly only because of the limit
        if (r == 1) {
          const uint32_t w = sc.surf[cell];
          nrm = surf_normal<M>(w, smooth);
          if (LAYOUT >= 2) p = p * (1.0f / fres);  // (exact: back to the reference's units)
          const v3 hit = madv(p, ld3(o.voxelBounds2), -ld3(o.voxelBounds));
          const float d = length(rpos - hit) - o.voxelSize;
          if (d < rd) { rd = d; rc = band_of((int)(w & 0xffu)); }
        }
      } else
      while (--steps >= 0) {
        int qx, qy, qz;
        M::cell3(p.x * frx, p.y * fry, p.z * frz, qx, qy, qz);
        if (COUNT) cnt.march_steps++;
        if (!in_grid(qx, qy, qz)) break;
        if (COUNT) cnt.vox_reads++;
        const int v = sc.vox[qz * o.voxelRes[3] + qy * o.voxelRes[0] + qx];
        if (v > iso) {
          nrm = smooth ? smooth_gradient(qx, qy, qz) : normalize(cell_gradient(qx, qy, qz));
          const v3 hit = madv(p, ld3(o.voxelBounds2), -ld3(o.voxelBounds));
          const float d = length(rpos - hit) - o.voxelSize;
          if (d < rd) { rd = d; rc = v < 168 ? (v < 84 ? 1.0f : 2.0f) : 3.0f; }
          break;
        }
        p = p + delta;
      }
    }
    dist = rd;
    code = rc;
  }

  struct Hit { v3 pos, normal; float distance; int objectID; };

  // outer march: renderer.cl:239-257
  //
  // Most distance estimates of a ray never reach the voxel walk: the ray is past
  // the clip box, misses it, or the ground is closer than the box entry -- yet
  // the reference pays a slab test (6 IEEE divisions) for each of them.  For a
  // ray ro + t*rd the entry/exit parameters at distance t are (lo - t, hi - t)
  // with lo/hi evaluated ONCE at t = 0, so an approximate copy of them decides
  // the test whenever the outcome is not within `slack` of flipping; only the
  // ambiguous estimates (and those that do walk) run the exact code.  The
  // outcome of every skipped test is certain, so results do not change.
  struct BoxFilter {
    // thresholds on the march distance t (>= 0) derived once per ray from the approximate
    // entry / exit parameters near0 / far0 and their error bound `slack`; each comparison
    // below implies the corresponding statement about the exact slab test with the margin
    // m(t) = slack + 8e-6 |t| (t-proportional rounding of the reference's own evaluation)
    float past;    // t > past            =>  far0 - t < -m(t): the box is entirely behind
    float before;  // t + 8e-6|t| + g < before  =>  near0 - t > g + m(t): entry farther than the ground term
                   //   (a line that misses the box by a margin gets before = 64: true while t <= 64)
    // (Rounds 2-4 also kept in_lo / in_hi / g_min -- "t in [in_lo, in_hi) and g > g_min => the position is inside the
    //  box, the slab test returns +0" -- to skip the six divisions of the slab test.  Three more floats live across
    //  every walk of the ray: without them the frame kernel spills 5-23 VGPRs fewer and config 2 / 3 / 5 are 3 / 2 /
    //  1-4 % faster in every contract, round 5; the margin test inside scene_distance still catches those estimates.)
  };
  RM_DEV BoxFilter make_filter(v3 ro, v3 rd) {
    const RmOpts& o = *sc.o;
    BoxFilter f;
    const float ax = __builtin_fabsf(rd.x), ay = __builtin_fabsf(rd.y), az = __builtin_fabsf(rd.z);
    // tiny components make the parameters huge (and 0 makes them inf/NaN): no filter
    const bool ok = fminf(fminf(ax, ay), az) >= 1e-3f && fmaxf(fmaxf(ax, ay), az) <= 2.0f &&
                    fmaxf(fmaxf(__builtin_fabsf(ro.x), __builtin_fabsf(ro.y)), __builtin_fabsf(ro.z)) <= 64.0f &&
                    o.startDist >= 0.0f && o.startDist <= 64.0f;
    const float ix = __builtin_amdgcn_rcpf(rd.x), iy = __builtin_amdgcn_rcpf(rd.y),
                iz = __builtin_amdgcn_rcpf(rd.z);
    const float lx = (o.voxelBoundsMin[0] - ro.x) * ix, hx = (o.voxelBoundsMax[0] - ro.x) * ix;
    const float ly = (o.voxelBoundsMin[1] - ro.y) * iy, hy = (o.voxelBoundsMax[1] - ro.y) * iy;
    const float lz = (o.voxelBoundsMin[2] - ro.z) * iz, hz = (o.voxelBoundsMax[2] - ro.z) * iz;
    const float near0 = fmaxf(fmaxf(fminf(lx, hx), fminf(ly, hy)), fminf(lz, hz));
    const float far0 = fminf(fminf(fmaxf(lx, hx), fmaxf(ly, hy)), fmaxf(lz, hz));
    // positions are rounded to ~4e-6 and divided by >= 1e-3, quotients (< 7e4) to ~8e-3
    const float slack = 0.03f + 8e-6f * (__builtin_fabsf(near0) + __builtin_fabsf(far0));
    // t(1 - 8e-6) > far0 + slack, with t >= 0
    f.past = fmaxf(far0 + slack, 0.0f) * 1.00002f;
    // the line misses the box (b < a) by more than m(64): no walk anywhere up to t = 64
    const bool miss = far0 - near0 < -(slack + 6e-4f);
    f.before = miss ? 64.0f : near0 - slack * 1.01f;
    if (!ok) {
      f.past = __builtin_inff();
      f.before = -__builtin_inff();
    }
    return f;
  }
  // true when the estimate at distance t certainly returns the ground/sky term
  // (renderer.cl:214 condition false) -- ground distance `g` = res.x there
  RM_DEV bool surely_no_walk(const BoxFilter& f, float t, float g) {
    // (straight-line on purpose: this sits in the hottest loop of the march)
    return (g <= 0.0f)                                               // entry distance is >= 0 or -1: never < g
           | (t > f.past)                                            // box entirely behind
           | (__builtin_fmaf(__builtin_fabsf(t), 8e-6f, t) + g < f.before);  // entry farther than the ground term / box missed
  }
  // distance_only: the caller uses r.distance alone (shadow rays): walks may stop where a hit
  // could no longer change it (walk_limit_for)
  // unit_dir: rdir comes straight out of normalize() (camera rays, light directions): its length is 1 to a few ulps, so
  // the walk limits -- bounds whose direction of error alone matters -- take 0.9999 for it and the samples per unit
  // become a launch-uniform number held in a scalar register (reflected directions are not unit vectors: measured there).
  // (normalize() of the contract's library returns a unit vector for EVERY finite non-zero input -- it rescales inputs whose
  //  squared length would underflow or overflow --; for a zero or NaN input it returns zero / NaN, and then every sample of
  //  a walk lies in the cell of the first one (or nowhere): a walk cut short ends as the whole one would.  Lights inside
  //  the volume, on the ground plane and at 1e30: tests/test_gpu_parity.py test_walk_limits_with_unusual_records.)
  RM_DEV void march(v3 ro, v3 rdir, Hit& r, float maxDist, int maxSteps, bool smooth,
                    bool distance_only = false, bool unit_dir = false) {
    const RmOpts& o = *sc.o;
    if (COUNT) cnt.rays++;
    float dist = o.startDist;
    // (the filter reasons about the clip box of the byte grid: off for the counting variant,
    //  which must run the plain algorithm, and for the quality mode, whose field extends
    //  beyond the box)
    constexpr bool kFilter = !COUNT && !SDFM;
    BoxFilter flt{};
    if (kFilter) flt = make_filter(ro, rdir);
    // Only the LAST estimate's position, code and normal survive the loop
    // (renderer.cl:244-246 overwrite them every turn), so the filtered turns -- the
    // large majority -- just remember that they were last; `last_t` is the distance
    // the surviving position was computed at.
    float last_t = dist, scode = 0.0f;
    // Two nested loops instead of the reference's one: the inner loop runs a lane
    // through consecutive turns whose estimate is certainly the ground / sky term
    // (cheap, ~85 % of all turns) WITHOUT waiting for the other lanes; the outer
    // loop then lets every lane that reached a turn needing the real estimate (slab
    // test + voxel walk) take it together.  With a single loop the wavefront paid a
    // full walk in almost every turn because some lane always needed one; now it
    // pays one per round, and a ray has only 2-3 such turns.  Per lane the sequence
    // of operations is unchanged.
    // (Loop state is one int per lane, not several bools: the compiler keeps bools
    // that live across a divergent loop as lane masks and spends three scalar
    // instructions per flag and turn on them.)
    //   why: 1 = this turn needs the real estimate, 2 = out of turns,
    //        3 = stopped in a filtered turn, 4 = stopped in an estimated turn
    const int turns0 = maxSteps;
    bool cut_last = false;
    const float dir_len = (ACCEL && !COUNT) ? (unit_dir ? 0.9999f : __builtin_amdgcn_sqrtf(dot(rdir, rdir)) * 0.9999f) : 1.0f;
    float spu = (ACCEL && !COUNT) ? samples_per_unit(o.maxVoxelIter, dir_len) : 0.0f;
    if (ACCEL && !COUNT && unit_dir) spu = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(spu)));
    int why;
    int last_kind = 0;  // 1: the last executed turn took the real estimate
    for (;;) {
      float g = 0.0f;
      why = 2;
      if (maxSteps > 0) {
        // one exit: a turn either continues (filtered, not converged, turns left) or not
        bool nw, go;
        do {
          maxSteps--;
          last_t = dist;
          const float h = M::fuse(rdir.y, dist, ro.y) + o.groundY;  // y of renderer.cl:244, then :211
          g = h < 1e5f ? h : 1e5f;
          nw = kFilter && surely_no_walk(flt, dist, g);
          go = nw & !((__builtin_fabsf(g) <= o.eps) | (dist >= maxDist));
          dist = go ? dist + g : dist;
        } while (go & (maxSteps > 0));
        why = go ? 2 : (nw ? 3 : 1);
        if (nw) last_kind = 0;
      }
      if (why != 1) break;
      float sd;
      int limit = 0x7fffffff;
      // (`dist` counts in units of |rdir|: the remaining world distance is (maxDist - dist) * |rdir|)
      if (ACCEL && distance_only)
        limit = walk_limit_from(fminf(g, ((maxDist - dist) + o.eps) * fmaxf(dir_len, 1.0f) * 1.001f), spu);
      // Marches whose hit is used keep distance and code of a turn when a hit lies beyond the
      // ground term, but take its NORMAL (renderer.cl:227-229 sets it before the union): only
      // the last turn's normal survives, so walk as far as the ground term now and repeat the
      // last turn without limit afterwards if its walk was cut (rare: the last turn of a ray
      // is a filtered one or finds its hit close by)
      if (ACCEL && !distance_only) limit = walk_limit_from(g, spu);
      cut_last = false;
      scene_distance(M::fuse3(rdir, dist, ro), rdir, o.maxVoxelIter, smooth, sd, scode, r.normal, false, limit,
                     &cut_last);
      last_kind = 1;
      if (__builtin_fabsf(sd) <= o.eps || dist >= maxDist) { why = 4; break; }
      dist += sd;
    }
    if (maxSteps != turns0) {  // at least one turn: renderer.cl:244-246 values of the last one
      r.pos = M::fuse3(rdir, last_t, ro);
      if (ACCEL && !distance_only && last_kind == 1 && cut_last) {
        float sd2, sc2;
        scene_distance(r.pos, rdir, o.maxVoxelIter, smooth, sd2, sc2, r.normal);
      }
      if (SDFM && last_kind == 1 && cut_last) r.normal = sdf_gradient(r.pos);  // (r.pos is the last estimate's position, bit for bit)
      if (last_kind == 0) {  // renderer.cl:211-212 for the ground / sky term
        const float h = M::fuse(rdir.y, last_t, ro.y) + o.groundY;
        scode = h < 1e5f ? h : -1.0f;
        r.normal = (h < 1e5f) ? V(0.f, 1.f, 0.f) : -rdir;
      }
      r.objectID = M::to_int(scode);
    }
    if (dist >= maxDist) {
      r.pos = M::fuse3(rdir, dist, ro);
      r.objectID = -1;
      dist = 1000.0f;
    }
    r.distance = dist;
  }


  RM_DEV v3 sky(v3 dir) { return sky_of<M>(*sc.o, dir); }

  // lseed: the seed of the light jitter (renderer.cl:267), a function of px, py and time alone: computed once, so that
  // px and py are dead after the camera ray
  struct Sample { v3 eye; v3 mcNormal; float px, py; float time; uint32_t lseed; };

  // jittered light position: renderer.cl:263-269
  RM_DEV v3 light_at(const Sample& s, int i) {
    const RmOpts& o = *sc.o;
    const float4 r = table(s.lseed);
    return mads(V(r.x, r.y, r.z), o.lightScatter, ld3(o.lightPos[i]));
  }
  RM_DEV v3 reflect(v3 v, v3 n) { return reflect_of<M>(v, n); }
  // fog + flares: renderer.cl:275-290
  RM_DEV v3 atmosphere(const Sample& s, v3 ro, v3 rdir, float dist, v3 col) {
    const RmOpts& o = *sc.o;
    const float fa = 1.0f - M::exp(dist * dist * -o.fogPow);
    const v3 sk = sky(rdir);
    col = mads(sk - col, fa, col);
    const int nl = o.numLights;
    for (int i = 0; i < nl; i++) {
      v3 lp = light_at(s, i);
      const float d = M::clamp(dot(lp - ro, rdir), 0.0f, dist);
      lp = mads(rdir, d, ro - lp);
      const float k = M::div(o.flareAmp, dot(lp, lp));
      col = mads(ld3(o.lightColor[i]), k, col);
    }
    return col;
  }
  // renderer.cl:292-301
  RM_DEV float shadow_term(v3 p, v3 ldir, float lmax) {
    Hit h{};
    march(p, ldir, h, lmax, sc.o->shadowIter, false, true, true);
    return M::step(lmax, h.distance);
  }
  RM_DEV float schlick(float r0, float smooth, v3 n, v3 view) { return schlick_of<M>(r0, smooth, n, view); }
  RM_DEV float blinn_phong(float smooth, v3 raydir, v3 ldir, v3 n) {
    return blinn_phong_of<M>(smooth, raydir, ldir, n);
  }
  // renderer.cl:327-346
  RM_DEV float occlusion(const Sample& s, v3 pos, v3 normal) {
    const RmOpts& o = *sc.o;
    if (COUNT) cnt.ao_calls++;
    float ao = 1.0f;
    float d = 0.0f;
    uint32_t seed =
        seed_of(M::fuse(s.time, 2671.918f, M::fuse(pos.z, 2945.87f, M::fuse(pos.x, 3183.75f, pos.y * 1831.42f))));
    for (int i = 0; i <= o.aoIter && (double)ao > 0.01; i++) {
      d += o.aoStepDist;
      seed += 37u;
      const float4 r = table(seed);
      const v3 n = normalize(mads(V(r.x, r.y, r.z), 0.2f, normal));
      float sd, scode;
      v3 nn;
      const v3 rpos = mads(n, d, pos);
      scene_distance(rpos, n, o.maxVoxelIter / 2, false, sd, scode, nn, false,
                     ACCEL ? ao_walk_limit(d, rpos.y + o.groundY, o.maxVoxelIter / 2) : 0x7fffffff);
      ao *= 1.0f - M::fmax(M::div((d - sd) * o.aoAmp, d), 0.0f);
    }
    return ao;
  }
  // renderer.cl:348-381
  RM_DEV v3 lighting(const Sample& s, v3 raydir, v3 hitpos, const Material& m, v3 normal,
                     v3 reflectCol) {
    const RmOpts& o = *sc.o;
    const float ao = occlusion(s, hitpos, normal);
    v3 diff = sky(normal) * ao;
    v3 spec = reflectCol * ao;
    v3 out = V(0.f, 0.f, 0.f);
    const int nl = o.numLights;
    for (int i = 0; i < nl; i++) {
      const v3 dl = light_at(s, i) - hitpos;
      const float d2 = dot(dl, dl);
      const float att = M::inv(d2);
      if (att > o.minLightAtt) {
        const v3 ldir = normalize(dl);
        const float lmax = M::fmin(M::sqrt(d2) - o.shadowBias, o.maxDist);
        const float sh = SDFM ? soft_shadow_sdf(M::fuse3(ldir, o.shadowBias, hitpos), ldir, lmax)
                              : shadow_term(M::fuse3(ldir, o.shadowBias, hitpos), ldir, lmax);
        if (sh > 0.0f) {
          const v3 inc = (ld3(o.lightColor[i]) * sh) * att;
          diff = M::fuse3(inc, M::fmax(0.0f, dot(ldir, normal)), diff);  // diffReflect += intensity * incidentLight
          spec = M::fuse3(inc, blinn_phong(m.smoothness, raydir, ldir, normal), spec);
        }
      }
      diff = diff * m.albedo;
      out = out + mixs(diff, spec, schlick(m.r0, m.smoothness, normal, raydir));
    }
    const float fl = (float)nl;
    return V(M::div(out.x, fl), M::div(out.y, fl), M::div(out.z, fl));
  }
  // one reflection bounce: renderer.cl:383-405
  RM_DEV v3 bounce_colour(const Sample& s, v3 ro, v3 rdir, Hit& h) {
    const RmOpts& o = *sc.o;
    march(ro, rdir, h, o.maxDist, o.maxIter, false);
    v3 col;
    if (h.objectID < 0) {
      col = sky(rdir);
    } else {
      const Material m = material(h.objectID);
      col = lighting(s, rdir, h.pos, m, h.normal, sky(reflect(rdir, h.normal)));
    }
    return atmosphere(s, ro, rdir, h.distance, col);
  }
  // primary shading: renderer.cl:407-446
  RM_DEV v3 sample_colour(const Sample& s, v3 ro, v3 rdir) {
    const RmOpts& o = *sc.o;
    Hit h{};
    march(ro, rdir, h, o.maxDist, o.maxIter, true, false, true);
    v3 col;
    if (h.distance >= o.maxDist) {
      col = sky(rdir);
    } else {
      if (COUNT) cnt.primary_hits++;
      const Material m = material(h.objectID);
      const float k = M::inv(M::mad(m.smoothness, 200.0f, 5.0f));
      const v3 norm = mads(s.mcNormal, k, h.normal);
      v3 refl = V(0.f, 0.f, 0.f);
      if (m.r0 > 0.0f && o.reflectIter > 0) {
        Hit rh{};
        rh.pos = h.pos;
        rh.normal = norm;
        v3 dir = rdir;
        for (int i = 0; i < o.reflectIter; i++) {
          dir = reflect(dir, rh.normal);
          const v3 from = M::fuse3(dir, 0.0075f, rh.pos);
          refl = refl + bounce_colour(s, from, dir, rh);
          if (rh.objectID < 0) break;
          if ((double)material(rh.objectID).r0 < 0.001) break;
        }
      } else {
        refl = sky(reflect(rdir, norm));
      }
      col = lighting(s, rdir, h.pos, m, norm, refl);
    }
    return atmosphere(s, ro, rdir, h.distance, col);
  }

  // renderer.cl:467-476 (sample state) and :456-465 (camera ray)
  RM_DEV Sample sample_init(int id) {
    const RmOpts& o = *sc.o;
    Sample s;
    s.time = time_;
    const int resx = o.resolution[0];
    const float fx = (float)(id % resx), fy = (float)(id / resx);
    const float4 mcPos = table((uint32_t)id * 17u + seed_of(time_ * 3141.3862f));
    const float4 t = table((uint32_t)id * 37u + seed_of(time_ * 1859.1467f));
    s.mcNormal = normalize(V(t.x, t.y, t.z));
    s.px = fx + mcPos.z;
    s.py = fy + mcPos.w;
    s.lseed = seed_of(M::fuse(s.time, 4763.742f, M::fuse(s.px, 1957.0f, s.py * 2173.0f)));
    s.eye = mads(V(s.mcNormal.z, s.mcNormal.x, s.mcNormal.y), o.dof, ld3(o.eyePos));
    return s;
  }
  RM_DEV v3 camera_dir(const Sample& s) {
    const RmOpts& o = *sc.o;
    const v3 fwd = normalize(ld3(o.targetPos) - s.eye);
    const v3 right = normalize(cross(fwd, ld3(o.up)));
    float vx = M::fuse(M::div(s.px, (float)o.resolution[0]), o.fov, -(o.fov * 0.5f));  // pixelPos / res * fov - fov * 0.5f
    float vy = M::fuse(M::div(s.py, (float)o.resolution[1]), o.fov, -(o.fov * 0.5f));
    vy *= -o.invAspect;
    const v3 upv = cross(right, fwd);
    return normalize(M::fuse3(right, vx, upv * vy) + fwd);  // (right * x + up * y) + forward
  }

  // =====================================================================================
  // The same sample with its secondary rays SHARED by the wavefront.
  //
  // In sample_colour() the AO probes and shadow marches of a hit run in the lane that owns the
  // hit while lanes without one (sky, or done earlier) idle: 25 % / 37 % of the lane slots of
  // those walks do work.  Every such ray is a pure function of a few floats of its owner
  // (position, normal, seed / light jitter), so ANY lane can trace it: the owners post their
  // inputs in LDS, the (owner, probe) and (owner, light) tasks are dealt round-robin to all
  // lanes of the wavefront, results come back through LDS and the owners combine them in the
  // reference's order.  Per task the operations are the ones of occlusion() / lighting(), so
  // the bits do not change.  Control flow around the shared phases is wave-uniform (decided by
  // ballots), which is what lets idle lanes take part.
  // =====================================================================================
  static constexpr int kWaveLdsIn = 9;    // posted inputs per lane
  static constexpr int kWaveLdsRes = RM_WAVE_AO_PROBES;   // results per lane (AO probes <= 8, lights <= 4)
  static_assert(kWaveLdsRes >= 4, "the shadow phase keeps one result per light (numLights <= 4)");
  static constexpr int kWaveLdsFloats = (kWaveLdsIn + kWaveLdsRes + 1) * 64;
  RM_DEV float& lds_in(int f, int lane) { return lds_[f * 64 + lane]; }
  RM_DEV float& lds_res(int f, int lane) { return lds_[(kWaveLdsIn + f) * 64 + lane]; }
  RM_DEV int& lds_map(int rank) { return reinterpret_cast<int*>(lds_)[(kWaveLdsIn + kWaveLdsRes) * 64 + rank]; }
  RM_DEV static void wave_sync() { __syncthreads(); }  // one wavefront per workgroup: orders its LDS traffic

  // bit patterns +0 .. +inf (no sign bit, not NaN) / +0 .. largest finite
  RM_DEV static bool sign_clear(v3 a) {
    return (__float_as_uint(a.x) <= 0x7f800000u) & (__float_as_uint(a.y) <= 0x7f800000u) &
           (__float_as_uint(a.z) <= 0x7f800000u);
  }
  RM_DEV static bool finite_nonneg(v3 a) {
    return (__float_as_uint(a.x) < 0x7f800000u) & (__float_as_uint(a.y) < 0x7f800000u) &
           (__float_as_uint(a.z) < 0x7f800000u);
  }
  struct Deal {  // who does what in a shared phase
    int lane, helpers, my_slot, owners, my_rank;
  };
  RM_DEV Deal deal(bool active) {
    Deal dl;
    const unsigned long long here = __ballot(1), own = __ballot(active);
    dl.lane = (int)(threadIdx.x & 63);
    const unsigned long long below = (1ull << dl.lane) - 1ull;
    dl.helpers = __popcll(here);
    dl.my_slot = __popcll(here & below);
    dl.owners = __popcll(own);
    dl.my_rank = __popcll(own & below);
    return dl;
  }
  // t / a and t % a for 0 <= t < 1024, 1 <= a <= 64 without an integer division
  RM_DEV static void divmod_small(int t, int a, int& q, int& r) {
    q = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)a));
    r = t - q * a;
    if (r < 0) { q--; r += a; }
    if (r >= a) { q++; r -= a; }
  }

  // occlusion() for all lanes of the wavefront at once; `active` lanes own a hit
  //
  // The exchange area has kWaveLdsRes result slots per owner.  kChunkedAO (the table layouts with the grid edge compiled
  // in: 3, 4, 5 -- the ones BASELINE's configurations use): a record that asks for more probes gets them in CHUNKS of that
  // many -- the owners fold a chunk's results into their product (renderer.cl:338-344, in probe order, with the reference's
  // early exit) before the next chunk overwrites them --, so any aoIter goes through the frame kernel.  The generic layouts
  // (0, 1, 2) keep one chunk: three more launch-uniform values tip their instantiations over the scalar register file
  // (2-4 spilled SGPRs, refused by the build's lint, round 5), and the host sends their frames with more than
  // kWaveLdsRes probes through the single-pass kernels (rm_api.hip frame_on_device, rmk::frame_takes_any_ao).
  static constexpr bool kChunkedAO = fixed_log2(LAYOUT) != 0;
  RM_DEV float occlusion_wave(bool active, const Sample& s, v3 pos, v3 normal) {
    if constexpr (kChunkedAO) return occlusion_wave_chunks(active, s, pos, normal);
    else return occlusion_wave_one_chunk(active, s, pos, normal);
  }
  // the generic layouts' form: ONE chunk (spelled out on its own: the chunk loop below with a constant trip count still costs
  // their instantiations 2 spilled SGPRs)
  RM_DEV float occlusion_wave_one_chunk(bool active, const Sample& s, v3 pos, v3 normal) {
    const RmOpts& o = *sc.o;
    // (aoIter + 1 <= kWaveLdsRes: the host sends frames whose records ask for more probes through the single-pass kernels.
    //  The clamp keeps a record rewritten in place behind the host's validation from writing past the exchange area:
    //  such a frame gets too few probes, never corrupted LDS.)
    const int np = min(o.aoIter + 1, kWaveLdsRes);
    const Deal dl = deal(active);
    if (dl.owners == 0) return 1.0f;
    const uint32_t seed0 =
        seed_of(M::fuse(s.time, 2671.918f, M::fuse(pos.z, 2945.87f, M::fuse(pos.x, 3183.75f, pos.y * 1831.42f))));
    if (active) {
      lds_in(0, dl.lane) = pos.x; lds_in(1, dl.lane) = pos.y; lds_in(2, dl.lane) = pos.z;
      lds_in(3, dl.lane) = normal.x; lds_in(4, dl.lane) = normal.y; lds_in(5, dl.lane) = normal.z;
      lds_in(6, dl.lane) = __uint_as_float(seed0);
      const unsigned long long tp = (unsigned long long)mc_;
      lds_in(7, dl.lane) = __uint_as_float((uint32_t)tp);
      lds_in(8, dl.lane) = __uint_as_float((uint32_t)(tp >> 32));
      lds_map(dl.my_rank) = dl.lane;
    }
    wave_sync();
    const int tasks = np * dl.owners;
    for (int base = 0; base < tasks; base += dl.helpers) {  // uniform trip count
      const int t = base + dl.my_slot;
      if (t < tasks) {
        int probe, rank;
        divmod_small(t, dl.owners, probe, rank);  // probe-major: a round holds probes of one distance
        const int owner = lds_map(rank);
        const v3 opos = V(lds_in(0, owner), lds_in(1, owner), lds_in(2, owner));
        const v3 onrm = V(lds_in(3, owner), lds_in(4, owner), lds_in(5, owner));
        const uint32_t seed = __float_as_uint(lds_in(6, owner)) + 37u * (uint32_t)(probe + 1);
        const float4* tab = reinterpret_cast<const float4*>(
            (unsigned long long)__float_as_uint(lds_in(7, owner)) |
            ((unsigned long long)__float_as_uint(lds_in(8, owner)) << 32));
        float d = 0.0f, dj = 0.0f;  // d of probe i = i+1 sequential adds (renderer.cl:339)
        for (int j = 0; j < np; j++) {
          dj += o.aoStepDist;
          if (j == probe) d = dj;
        }
        const float4 r = tab[seed & (RM_TABLE_ENTRIES - 1)];
        const v3 n = normalize(mads(V(r.x, r.y, r.z), 0.2f, onrm));
        float sd, scode;
        v3 nn;
        const v3 rpos = mads(n, d, opos);
        const int ao_limit = ao_walk_limit(d, rpos.y + o.groundY, o.maxVoxelIter / 2);
        scene_distance(rpos, n, o.maxVoxelIter / 2, false, sd, scode, nn, false, ao_limit);
        lds_res(probe, owner) = sd;
      }
    }
    wave_sync();
    float ao = 1.0f;
    if (active) {
      float d = 0.0f;
      for (int i = 0; i < np && (double)ao > 0.01; i++) {  // renderer.cl:338-344
        d += o.aoStepDist;
        ao *= 1.0f - M::fmax(M::div((d - lds_res(i, dl.lane)) * o.aoAmp, d), 0.0f);
      }
    }
    wave_sync();  // the posted values are dead: the next shared phase may overwrite them
    return ao;
  }

  RM_DEV float occlusion_wave_chunks(bool active, const Sample& s, v3 pos, v3 normal) {
    const RmOpts& o = *sc.o;
    const int np_all = o.aoIter + 1;
    const Deal dl = deal(active);
    if (dl.owners == 0) return 1.0f;
    const uint32_t seed0 =
        seed_of(M::fuse(s.time, 2671.918f, M::fuse(pos.z, 2945.87f, M::fuse(pos.x, 3183.75f, pos.y * 1831.42f))));
    if (active) {
      lds_in(0, dl.lane) = pos.x; lds_in(1, dl.lane) = pos.y; lds_in(2, dl.lane) = pos.z;
      lds_in(3, dl.lane) = normal.x; lds_in(4, dl.lane) = normal.y; lds_in(5, dl.lane) = normal.z;
      lds_in(6, dl.lane) = __uint_as_float(seed0);
      const unsigned long long tp = (unsigned long long)mc_;
      lds_in(7, dl.lane) = __uint_as_float((uint32_t)tp);
      lds_in(8, dl.lane) = __uint_as_float((uint32_t)(tp >> 32));
      lds_map(dl.my_rank) = dl.lane;
    }
    wave_sync();
    float ao = 1.0f, dsum = 0.0f;  // the owner's running product and probe distance (renderer.cl:336-339)
    for (int p0 = 0; p0 < np_all; p0 += kWaveLdsRes) {  // uniform; ONE trip unless aoIter > 7
      const int np = min(np_all - p0, kWaveLdsRes);
      const int tasks = np * dl.owners;
      for (int base = 0; base < tasks; base += dl.helpers) {  // uniform trip count
        const int t = base + dl.my_slot;
        if (t < tasks) {
          int probe, rank;
          divmod_small(t, dl.owners, probe, rank);  // probe-major: a round holds probes of one distance
          const int owner = lds_map(rank);
          const v3 opos = V(lds_in(0, owner), lds_in(1, owner), lds_in(2, owner));
          const v3 onrm = V(lds_in(3, owner), lds_in(4, owner), lds_in(5, owner));
          const int gp = p0 + probe;  // the probe's number in the record's loop
          const uint32_t seed = __float_as_uint(lds_in(6, owner)) + 37u * (uint32_t)(gp + 1);
          const float4* tab = reinterpret_cast<const float4*>(
              (unsigned long long)__float_as_uint(lds_in(7, owner)) |
              ((unsigned long long)__float_as_uint(lds_in(8, owner)) << 32));
          float d = 0.0f, dj = 0.0f;  // d of probe i = i+1 sequential adds (renderer.cl:339)
          for (int j = 0; j < np_all; j++) {
            dj += o.aoStepDist;
            if (j == gp) d = dj;
          }
          const float4 r = tab[seed & (RM_TABLE_ENTRIES - 1)];
          const v3 n = normalize(mads(V(r.x, r.y, r.z), 0.2f, onrm));
          float sd, scode;
          v3 nn;
          const v3 rpos = mads(n, d, opos);
          const int ao_limit = ao_walk_limit(d, rpos.y + o.groundY, o.maxVoxelIter / 2);
          scene_distance(rpos, n, o.maxVoxelIter / 2, false, sd, scode, nn, false, ao_limit);
          lds_res(probe, owner) = sd;
        }
      }
      wave_sync();
      if (active) {
        for (int i = 0; i < np && (double)ao > 0.01; i++) {  // renderer.cl:338-344 (a product that has stopped stays stopped)
          dsum += o.aoStepDist;
          ao *= 1.0f - M::fmax(M::div((dsum - lds_res(i, dl.lane)) * o.aoAmp, dsum), 0.0f);
        }
      }
      wave_sync();  // the results are consumed: the next chunk / the next shared phase may overwrite them
    }
    return ao;
  }

  // the shadow marches of lighting() for all lanes: distance reached by the march towards
  // light i in lds_res(i, lane) (only where the light passes the attenuation test)
  // `need`: bit i set = this lane owns a hit whose light i needs its shadow march (lighting_wave)
  RM_DEV void shadows_wave(unsigned int need, v3 hitpos, v3 jit) {
    const RmOpts& o = *sc.o;
    const int nl = o.numLights;
    const bool active = need != 0u;
    const Deal dl = deal(active);
    if (dl.owners == 0 || nl <= 0) return;
    if (active) {
      lds_in(0, dl.lane) = hitpos.x; lds_in(1, dl.lane) = hitpos.y; lds_in(2, dl.lane) = hitpos.z;
      lds_in(3, dl.lane) = jit.x; lds_in(4, dl.lane) = jit.y; lds_in(5, dl.lane) = jit.z;
    }
    // the task list, light-major: one byte (light << 6 | owner lane) per needed (owner, light) pair
    uint8_t* const task_of = reinterpret_cast<uint8_t*>(&lds_map(0));
    int tasks = 0;
    for (int i = 0; i < nl && i < 4; i++) {  // uniform
      const bool mine = (need >> i) & 1u;
      const unsigned long long mk = __ballot(mine);
      if (mine) task_of[tasks + __popcll(mk & ((1ull << dl.lane) - 1ull))] = (uint8_t)((i << 6) | dl.lane);
      tasks += __popcll(mk);
    }
    wave_sync();
    for (int base = 0; base < tasks; base += dl.helpers) {
      const int t = base + dl.my_slot;
      if (t < tasks) {
        const int e = task_of[t];
        const int light = e >> 6, owner = e & 63;
        const v3 opos = V(lds_in(0, owner), lds_in(1, owner), lds_in(2, owner));
        const v3 ojit = V(lds_in(3, owner), lds_in(4, owner), lds_in(5, owner));
        // the expressions of lighting() (renderer.cl:356-362)
        const v3 dlv = mads(ojit, o.lightScatter, ld3(o.lightPos[light])) - opos;
        const float d2 = dot(dlv, dlv);
        const float att = M::inv(d2);
        if (att > o.minLightAtt) {
          const v3 ldir = normalize(dlv);
          const float lmax = M::fmin(M::sqrt(d2) - o.shadowBias, o.maxDist);
          if (SDFM) {  // quality mode: the penumbra estimate itself is the task's result
            lds_res(light, owner) = soft_shadow_sdf(M::fuse3(ldir, o.shadowBias, opos), ldir, lmax);
          } else {
            Hit h{};
            march(M::fuse3(ldir, o.shadowBias, opos), ldir, h, lmax, o.shadowIter, false, true, true);
            lds_res(light, owner) = h.distance;
          }
        }
      }
    }
    wave_sync();
  }

  // lighting() with the rays of all hits of the wavefront traced together
  RM_DEV v3 lighting_wave(bool active, const Sample& s, v3 raydir, v3 hitpos, int objectID,
                          v3 normal, bool mirror_sky, v3 reflectCol) {
    const RmOpts& o = *sc.o;
    if (__ballot(active) == 0) return V(0.f, 0.f, 0.f);  // uniform
    const float ao = occlusion_wave(active, s, hitpos, normal);
    // light jitter: one table value for all lights (renderer.cl:263-269)
    v3 jit = V(0.f, 0.f, 0.f);
    if (active) {
      const float4 r = table(s.lseed);
      jit = V(r.x, r.y, r.z);
    }
    // Which (hit, light) pairs need their shadow march at all.  A pair whose diffuse and
    // specular factors are both exactly +0 (the light is behind the surface: max(0, l.n) = 0,
    // and the half vector too, so blinn_phong takes its `return 0`) adds lightColor*sh*att * 0
    // to the running sums (renderer.cl:368-372): +0 when that product is finite and not
    // negative, and x + (+0) == x bit for bit for every x that is not -0.  The running sums
    // start from sky*ao / reflectCol*ao and only ever see + (non-negative) and * albedo, so if
    // all of those have a clear sign bit (and are not NaN) they never hold -0: the shadow
    // term of such a pair cannot reach the result and its march -- about a quarter of all
    // shadow marches -- is not traced.  (4 lights at most, the size of the record's arrays.)
    unsigned int need = 0u;
    if (active) {
      const Material m = material(objectID);
      const v3 rc = mirror_sky ? sky(reflect(raydir, normal)) : reflectCol;
      const v3 d0 = sky(normal) * ao, s0 = rc * ao;
      bool clean = sign_clear(d0) & sign_clear(s0) & sign_clear(m.albedo);
      unsigned int dark = 0u;
      for (int i = 0; i < o.numLights && i < 4; i++) {
        const v3 dl = mads(jit, o.lightScatter, ld3(o.lightPos[i])) - hitpos;
        const float att = M::inv(dot(dl, dl));
        if (att > o.minLightAtt) {
          need |= 1u << i;
          const v3 ldir = normalize(dl);
          const v3 inc = ld3(o.lightColor[i]) * att;
          clean &= finite_nonneg(inc);
          const bool back = M::fmax(0.0f, dot(ldir, normal)) == 0.0f;
          if (back && !(dot(normalize(ldir - raydir), normal) > 0.0f)) dark |= 1u << i;
        }
      }
      if (clean) need &= ~dark;
    }
    shadows_wave(need, hitpos, jit);
    v3 res = V(0.f, 0.f, 0.f);
    if (active) {
      const Material m = material(objectID);
      if (mirror_sky) reflectCol = sky(reflect(raydir, normal));
      v3 diff = sky(normal) * ao;
      v3 spec = reflectCol * ao;
      v3 out = V(0.f, 0.f, 0.f);
      const int nl = o.numLights;
      const int lane = (int)(threadIdx.x & 63);
      for (int i = 0; i < nl; i++) {
        const v3 dl = mads(jit, o.lightScatter, ld3(o.lightPos[i])) - hitpos;
        const float d2 = dot(dl, dl);
        const float att = M::inv(d2);
        if (att > o.minLightAtt) {
          const v3 ldir = normalize(dl);
          const float lmax = M::fmin(M::sqrt(d2) - o.shadowBias, o.maxDist);
          const float sh = SDFM ? lds_res(i, lane) : M::step(lmax, lds_res(i, lane));
          if (sh > 0.0f) {
            const v3 inc = (ld3(o.lightColor[i]) * sh) * att;
            diff = M::fuse3(inc, M::fmax(0.0f, dot(ldir, normal)), diff);  // diffReflect += intensity * incidentLight
            spec = M::fuse3(inc, blinn_phong(m.smoothness, raydir, ldir, normal), spec);
          }
        }
        diff = diff * m.albedo;
        out = out + mixs(diff, spec, schlick(m.r0, m.smoothness, normal, raydir));
      }
      const float fl = (float)nl;
      res = V(M::div(out.x, fl), M::div(out.y, fl), M::div(out.z, fl));
    }
    wave_sync();  // results consumed before the next shared phase posts
    return res;
  }

  // sample_colour() with wave-uniform control flow around the shared phases.  A lane that is
  // not `live` owns no sample in this turn (its pass lies beyond the frame's last): it traces
  // nothing of its own but still deals with the other lanes' secondary rays.
  RM_DEV v3 sample_colour_wave(const Sample& s, v3 ro, v3 rdir, bool live = true) {
    const RmOpts& o = *sc.o;
    Hit h{};
    if (live) march(ro, rdir, h, o.maxDist, o.maxIter, true, false, true);
    const bool hit = live && !(h.distance >= o.maxDist);
    v3 norm = V(0.f, 0.f, 0.f);
    float r0 = 0.0f;
    if (hit) {
      const Material m = material(h.objectID);
      const float k = M::inv(M::mad(m.smoothness, 200.0f, 5.0f));
      norm = mads(s.mcNormal, k, h.normal);
      r0 = m.r0;
    }
    const bool bounces = hit && r0 > 0.0f && o.reflectIter > 0;
    v3 refl = V(0.f, 0.f, 0.f);
    if (__ballot(bounces) != 0) {  // uniform
      Hit rh{};
      rh.pos = h.pos;
      rh.normal = norm;
      v3 dir = rdir;
      bool alive = bounces;
      for (int i = 0; i < o.reflectIter; i++) {  // uniform bound; lanes drop out through `alive`
        if (__ballot(alive) == 0) break;         // uniform
        v3 from = V(0.f, 0.f, 0.f);
        if (alive) {
          dir = reflect(dir, rh.normal);
          from = M::fuse3(dir, 0.0075f, rh.pos);
          march(from, dir, rh, o.maxDist, o.maxIter, false);  // bounce_colour(), renderer.cl:383-405
        }
        const bool bhit = alive && rh.objectID >= 0;
        const v3 lit = lighting_wave(bhit, s, dir, rh.pos, rh.objectID, rh.normal, true, V(0.f, 0.f, 0.f));
        if (alive) {
          const v3 col = bhit ? lit : sky(dir);
          refl = refl + atmosphere(s, from, dir, rh.distance, col);
          if (rh.objectID < 0) alive = false;
          else if ((double)material(rh.objectID).r0 < 0.001) alive = false;
        }
      }
    }
    if (hit && !bounces) refl = sky(reflect(rdir, norm));
    const v3 lit = lighting_wave(hit, s, rdir, h.pos, h.objectID, norm, false, refl);
    const v3 col = hit ? lit : sky(rdir);
    return atmosphere(s, ro, rdir, h.distance, col);
  }

  // shade() through the wave-shared path; every lane of the wavefront that has a pixel
  // must call it (lanes without one have left the kernel), lds = kWaveLdsFloats floats
  RM_DEV v3 shade_wave(int id, float* lds, bool live = true) {
    lds_ = lds;
    const Sample s = sample_init(id);
    const v3 rdir = camera_dir(s);
    return sample_colour_wave(s, s.eye, rdir, live) * sc.o->exposure;
  }

  // colour * exposure of work-item `id` (the value RenderImage blends in, renderer.cl:491)
  RM_DEV v3 shade(int id) {
    const Sample s = sample_init(id);
    const v3 rdir = camera_dir(s);
    return sample_colour(s, s.eye, rdir) * sc.o->exposure;
  }
};

}  // namespace rmk
